#!/bin/bash
# r5 call h: f32 weight gradient on the LDS-DMA ring (conv_wgrad_dma[_lin]_f32_kernel): parity, then cfg2 A/B through the experiments build
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5h; mkdir -p $O
rm -f gpurun_out/parity_report.txt
timeout 900 python -m pytest tests/test_round4_gpu.py tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -x -p no:cacheprovider --timeout 600 -k "wgrad or golden or train or step or grad" > $O/tests.log 2>&1; tail -4 $O/tests.log
grep "wgrad" gpurun_out/parity_report.txt | grep float32 | head -12
run() { L=$1; shift
  env "$@" timeout 300 python tools/bench_experiments.py --layers 50 --size 480 --batch 16 --dtype f32 --subnet keypoint_subnet --steps 15 --warmup 4 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['ms_per_step_median_hipevent'], d['value'])" | tee -a $O/cfg2_ab.txt
}
run generic MPN_WGRAD_F32_DMA=0
run dma X=1
run generic MPN_WGRAD_F32_DMA=0
run dma X=1
run dma_target512 MPN_WGRAD_TARGET=512
run dma_target1024 MPN_WGRAD_TARGET=1024
timeout 400 python bench.py --layers 50 --size 480 --batch 16 --dtype f32 --subnet keypoint_subnet --steps 20 --warmup 5 --no-cpu-baseline > $O/cfg2_bench.json 2> $O/cfg2.err
python -c "import json; d=json.loads(open('$O/cfg2_bench.json').read().strip().splitlines()[-1]); print('cfg2 production', d['value'], d['ms_per_step'], d.get('roofline',{}).get('kernel'), d.get('roofline',{}).get('frac')); print(d.get('kernel_classes_ms_per_step'))"
