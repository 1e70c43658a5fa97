#!/bin/bash
# round 4, call Q: wgrad fragment prefetch (MPN_WGRAD_PF=1): parity, phase profile, micro-benchmark, step A/B (slice target 500 keeps <= 512 workgroups)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4q; mkdir -p $O
MPN_WGRAD_PF=1 timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -p no:cacheprovider -k "wgrad or dgrad" > $O/tests_pf.log 2>&1; tail -3 $O/tests_pf.log
for v in 0 1; do
  echo "== MPN_WGRAD_PF=$v (target 500)"
  MPN_WGRAD_TARGET=500 MPN_WGRAD_PF=$v timeout 600 python tools/kloop_profile.py 2>&1 | grep "wgrad\|^[13]x" | sed 's/ | span.*, / | /'
done | tee $O/kloop_pf.txt
for v in 0 1 0 1; do
  echo "== microbench MPN_WGRAD_PF=$v"; MPN_WGRAD_TARGET=500 MPN_WGRAD_PF=$v MB_ONLY=2,3,4,5,6 MB_COLD=1 MB_ITERS=40 timeout 300 python tools/conv_microbench.py 2>&1 | grep "wgrad"
done | tee $O/microbench_pf.txt
ab() {  # label env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])"
}
for rep in 1 2 3; do
  ab pf0 MPN_WGRAD_TARGET=500 MPN_WGRAD_PF=0
  ab pf1 MPN_WGRAD_TARGET=500 MPN_WGRAD_PF=1
done 2>&1 | tee $O/step_ab.txt
