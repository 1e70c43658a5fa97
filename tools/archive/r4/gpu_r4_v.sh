#!/bin/bash
# round 4, call V: wgrad with a dedicated loader wave (MPN_WGRAD_LDR=3 / 4 ring slots): DMA phase and MFMA phase of one workgroup side by side
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4v; mkdir -p $O
for v in 3 4; do
MPN_WGRAD_LDR=$v timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -p no:cacheprovider -k "wgrad" > $O/tests_ldr$v.log 2>&1; tail -3 $O/tests_ldr$v.log
done
for v in 0 3 4; do
  echo "== MPN_WGRAD_LDR=$v"
  MPN_WGRAD_LDR=$v timeout 600 python tools/kloop_profile.py 2>&1 | grep "wgrad\|^[13]x" | sed 's/ | span.*, / | /'
done | tee $O/kloop_ldr.txt
for v in 0 3 4 0 3 4; do
  echo "== microbench MPN_WGRAD_LDR=$v"; MPN_WGRAD_LDR=$v MB_ONLY=2,3,4,5,6 MB_COLD=1 MB_ITERS=40 timeout 300 python tools/conv_microbench.py 2>&1 | grep "wgrad"
done | tee $O/microbench_ldr.txt
ab() {  # label env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])"
}
for rep in 1 2; do
  ab ldr0 MPN_WGRAD_LDR=0
  ab ldr3 MPN_WGRAD_LDR=3
  ab ldr4 MPN_WGRAD_LDR=4
done 2>&1 | tee $O/step_ab.txt
