#!/bin/bash
# in-call A/B of two library builds: tools/libmpn_old.so vs tools/libmpn_new.so (interleaved bench runs)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
L=multiposenet/pytorch_amd/libmpn_hip.so
for V in old new old new old new; do
  cp tools/libmpn_$V.so $L
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V', d['ms_per_step_median_hipevent'], d['ms_per_step'])"
done
for V in old new; do
  cp tools/libmpn_$V.so $L
  echo "microbench $V"; MB_ONLY=0,2,3,4 MB_WGRAD=0 MB_COLD=1 MB_ITERS=40 timeout 200 python tools/conv_microbench.py 2>&1 | grep -v "amdgpu\|DEBUG"
done
cp tools/libmpn_new.so $L
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q -x -m gpu 2>&1 | tail -2
