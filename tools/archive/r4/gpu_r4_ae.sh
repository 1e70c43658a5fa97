#!/bin/bash
# round 4, call AE: shared-tap 3x3 weight gradient (conv_wgrad_s3_kernel, MPN_WGRAD_S3 default 1): parity, k-loop profile, isolated launches, step
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4ae; mkdir -p $O
timeout 900 python -m pytest tests/test_round4_gpu.py -q -x -m gpu -p no:cacheprovider -k "instantiations" > $O/tests_s3_a.log 2>&1; tail -15 $O/tests_s3_a.log | cut -c1-300
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_round2_gpu.py tests/test_round3_gpu.py -q -x -m gpu -p no:cacheprovider -k "wgrad or backward or train or grad or pyramid" > $O/tests_s3_b.log 2>&1; tail -3 $O/tests_s3_b.log | cut -c1-300
for v in 0 1; do
  echo "== MPN_WGRAD_S3=$v"
  MPN_WGRAD_S3=$v timeout 600 python tools/kloop_profile.py 2>&1 | grep "wgrad\|^[13]x" | sed 's/ | span.*, / | /'
done | tee $O/kloop_s3.txt
for v in 0 1 0 1; do
  echo "== microbench MPN_WGRAD_S3=$v"; MPN_WGRAD_S3=$v MB_ONLY=4,5,7 MB_COLD=1 MB_ITERS=40 timeout 300 python tools/conv_microbench.py 2>&1 | grep "wgrad"
done | tee $O/microbench_s3.txt
ab() {  # label env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])"
}
for rep in 1 2 3; do
  ab s3_0 MPN_WGRAD_S3=0
  ab s3_1 MPN_WGRAD_S3=1
done 2>&1 | tee $O/step_ab.txt
