#!/bin/bash
# round 4, call AA: conv_igemm with 128-byte tile rows (MPN_IGEMM_K64=1: two k-steps per ring stage, whole-line DMA)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4aa; mkdir -p $O
MPN_IGEMM_K64=1 timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_round2_gpu.py tests/test_round3_gpu.py -q -x -m gpu -p no:cacheprovider -k "conv or dgrad or bottleneck or forward or train or fold" > $O/tests_k64.log 2>&1; tail -3 $O/tests_k64.log
for v in 0 1; do
  echo "== MPN_IGEMM_K64=$v"
  MPN_IGEMM_K64=$v timeout 600 python tools/kloop_profile.py 2>&1 | grep -v "wgrad" | sed 's/ | span.*, / | /'
done | tee $O/kloop_k64.txt
for v in 0 1 0 1; do
  echo "== microbench MPN_IGEMM_K64=$v"; MPN_IGEMM_K64=$v MB_ONLY=0,1,2,3 MB_COLD=1 MB_ITERS=40 MB_WGRAD=0 timeout 300 python tools/conv_microbench.py 2>&1 | grep -v wgrad
done | tee $O/microbench_k64.txt
ab() {  # label env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])"
}
for rep in 1 2 3; do
  ab k64_0 MPN_IGEMM_K64=0
  ab k64_1 MPN_IGEMM_K64=1
done 2>&1 | tee $O/step_ab.txt
