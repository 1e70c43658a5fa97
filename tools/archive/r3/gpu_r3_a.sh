#!/bin/bash
# round 3, first GPU call: new tests, harness, RCCL shared-GPU probe, baseline bench
mkdir -p gpurun_out/r3a
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_round3_gpu.py tests/test_harness_gpu.py tests/test_replay_gpu.py -x -q -m gpu > gpurun_out/r3a/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3a/tests.log
tail -5 gpurun_out/r3a/tests.log
timeout 200 python tools/rccl_shared_gpu_probe.py > gpurun_out/r3a/rccl_probe.log 2>&1
cat gpurun_out/r3a/rccl_probe.log | tail -3
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err
cat gpurun_out/r3a/bench.json | cut -c1-600
