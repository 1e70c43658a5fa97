#!/bin/bash
# round 4, call AH: the other configurations on the final build — cfg4 / cfg2 traces + PMC, cfg5 inference stages
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
bash tools/gpu_profile_config.sh r4_cfg4 "--layers 101 --size 800 --batch 8 --dtype bf16" > gpurun_out/r4_cfg4.log 2>&1; cut -c1-260 gpurun_out/r4_cfg4/bench.json
bash tools/gpu_profile_config.sh r4_cfg2 "--layers 50 --size 480 --batch 16 --dtype f32 --subnet keypoint_subnet" > gpurun_out/r4_cfg2.log 2>&1; cut -c1-260 gpurun_out/r4_cfg2/bench.json
mkdir -p gpurun_out/r4_cfg5; timeout 600 python tools/infer_bench.py > gpurun_out/r4_cfg5/infer_bench.txt 2>&1; tail -8 gpurun_out/r4_cfg5/infer_bench.txt | cut -c1-300
