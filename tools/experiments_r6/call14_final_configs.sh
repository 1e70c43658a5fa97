#!/bin/bash
# round-6 end artefacts, part 2: cfg2 / cfg4 (trace + PMC + bench line with roofline.traffic), cfg5, NMS worst case, MFMA / LDS utilisation
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp MPN_ROUND=6
bash tools/gpu_profile_config.sh r6cfg2 "--layers 50 --size 480 --batch 16 --dtype f32 --subnet keypoint_subnet" r06_cfg2_pmc_hbm_traffic.json > gpurun_out/r6cfg2.log 2>&1
cut -c1-300 gpurun_out/r6cfg2/bench.json
bash tools/gpu_profile_config.sh r6cfg4 "--layers 101 --size 800 --batch 8" r06_cfg4_pmc_hbm_traffic.json > gpurun_out/r6cfg4.log 2>&1
cut -c1-300 gpurun_out/r6cfg4/bench.json
O=gpurun_out/r6final; mkdir -p $O
timeout 600 python tools/infer_bench.py --iters 10 2>/dev/null > $O/cfg5_infer_bench.txt; cut -c1-160 $O/cfg5_infer_bench.txt
timeout 300 python tools/nms_microbench.py > $O/nms_microbench.txt 2>&1; tail -8 $O/nms_microbench.txt | cut -c1-200
bash tools/gpu_pmc_util.sh r6final > $O/pmcutil.log 2>&1; head -30 $O/pmc_utilisation.txt | cut -c1-200
