#!/bin/bash
# round 4, call R: wgrad 128x128 tile on EIGHT waves (MPN_WGRAD_NW=8): parity, phase profile, micro-benchmark, step A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4r; mkdir -p $O
MPN_WGRAD_NW=8 timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q -x -m gpu -p no:cacheprovider > $O/tests_nw8.log 2>&1; tail -3 $O/tests_nw8.log
for v in 4 8; do
  echo "== MPN_WGRAD_NW=$v"
  MPN_WGRAD_NW=$v timeout 600 python tools/kloop_profile.py 2>&1 | grep "wgrad\|^[13]x" | sed 's/ | span.*, / | /'
done | tee $O/kloop_nw8.txt
for v in 4 8 4 8; do
  echo "== microbench MPN_WGRAD_NW=$v"; MPN_WGRAD_NW=$v MB_ONLY=2,3,4,5,6 MB_COLD=1 MB_ITERS=40 timeout 300 python tools/conv_microbench.py 2>&1 | grep "wgrad"
done | tee $O/microbench_nw8.txt
ab() {  # label env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])"
}
for rep in 1 2 3; do
  ab nw4 MPN_WGRAD_NW=4
  ab nw8 MPN_WGRAD_NW=8
done 2>&1 | tee $O/step_ab.txt
ab nw8_t500 MPN_WGRAD_NW=8 MPN_WGRAD_TARGET=500 | tee -a $O/step_ab.txt
ab nw8_t768 MPN_WGRAD_NW=8 MPN_WGRAD_TARGET=768 | tee -a $O/step_ab.txt
