#!/bin/bash
# round 4, call Y: LIN weight gradient with the halo predicate of the NEXT k-step computed behind the DMA queue (under the MFMA phase)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4y; mkdir -p $O
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_round2_gpu.py tests/test_round3_gpu.py -q -x -m gpu -p no:cacheprovider -k "wgrad or backward or train or grad or pyramid" > $O/tests_lin.log 2>&1; tail -1 $O/tests_lin.log
for v in 0 1; do
  echo "== MPN_WGRAD_LIN=$v"
  MPN_WGRAD_LIN=$v timeout 600 python tools/kloop_profile.py 2>&1 | grep "wgrad\|^[13]x" | sed 's/ | span.*, / | /'
done | tee $O/kloop_lin.txt
for v in 0 1 0 1; do
  echo "== microbench MPN_WGRAD_LIN=$v"; MPN_WGRAD_LIN=$v MB_ONLY=2,3,4,5,6 MB_COLD=1 MB_ITERS=40 timeout 300 python tools/conv_microbench.py 2>&1 | grep "wgrad"
done | tee $O/microbench_lin.txt
ab() {  # label env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])"
}
for rep in 1 2 3; do
  ab lin0 MPN_WGRAD_LIN=0
  ab lin1 MPN_WGRAD_LIN=1
done 2>&1 | tee $O/step_ab.txt
