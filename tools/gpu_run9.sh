#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s10; mkdir -p $O
for F in 0 64 128 16 32; do
  MPN_DEBUG_FLAGS=$F MB_ONLY=2,3,4 MB_WGRAD=0 MB_COLD=1 MB_ITERS=40 timeout 200 python tools/conv_microbench.py 2>&1 | grep -v amdgpu | tee -a $O/ablate.txt
done
MB_ONLY=2,3,4 MB_WGRAD=1 MB_COLD=1 MB_ITERS=40 timeout 200 python tools/conv_microbench.py 2>&1 | grep -v amdgpu | tee -a $O/ablate.txt
