#!/bin/bash
# round 4, call G: wgrad with the DMA issue interleaved between the MFMAs (MPN_WGRAD_ILV=1): parity, phase profile, micro-benchmark, step A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4g; mkdir -p $O
MPN_WGRAD_ILV=1 timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -p no:cacheprovider -k "wgrad or dgrad" > $O/tests_ilv.log 2>&1; tail -3 $O/tests_ilv.log
for v in 0 1; do
  echo "== MPN_WGRAD_ILV=$v"
  MPN_WGRAD_ILV=$v timeout 600 python tools/kloop_profile.py 2>&1 | grep -A1 "wgrad\|^[13]x"
done | grep -v "^--" | tee $O/kloop_ilv.txt
for v in 0 1 0 1; do
  echo "== microbench MPN_WGRAD_ILV=$v"; MPN_WGRAD_ILV=$v MB_ONLY=2,3,4,5,6 MB_COLD=1 MB_ITERS=40 timeout 300 python tools/conv_microbench.py 2>&1 | grep "wgrad"
done | tee $O/microbench_ilv.txt
ab() {  # label env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])"
}
for rep in 1 2 3; do
  ab ilv0 MPN_WGRAD_ILV=0
  ab ilv1 MPN_WGRAD_ILV=1
done 2>&1 | tee $O/step_ab.txt
