#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s11; mkdir -p $O
timeout 600 python -m pytest tests/test_round2_gpu.py -q -x -m gpu -k "bn_finalize or bn_backward_statistics or folded" 2>&1 | tail -15 | tee $O/tests.txt
cp gpurun_out/parity_report.txt $O/ 2>/dev/null
MPN_BN_FUSED_FINALIZE=0 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events 2>&1 | tail -1 | tee $O/bench_fin0.json
MPN_BN_FUSED_FINALIZE=1 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events 2>&1 | tail -1 | tee $O/bench_fin1.json
timeout 600 python -m pytest tests/test_replay_gpu.py tests/test_model_gpu.py -q -x -m gpu 2>&1 | tail -5 | tee $O/tests2.txt
