#!/bin/bash
# round 4, call AJ: parity-class dgrads for f32 operands too (the fp32 parity path, cfg2): tests + cfg2 A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4aj; mkdir -p $O
timeout 900 python -m pytest tests/test_round4_gpu.py tests/test_kernels_gpu.py -q -x -m gpu -p no:cacheprovider -k "parity_class or stride2 or dgrad" > $O/tests_cls.log 2>&1; tail -6 $O/tests_cls.log | cut -c1-250
ab() {  # label env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events --layers 50 --size 480 --batch 16 --dtype f32 --subnet keypoint_subnet 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])"
}
for rep in 1 2; do
  ab f32_cls0 MPN_DGRAD_S2_CLASSES=0
  ab f32_cls1 MPN_DGRAD_S2_CLASSES=1
done 2>&1 | tee $O/step_ab_cfg2.txt
