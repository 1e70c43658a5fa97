#!/bin/bash
# round 4, call F: k-loop phase profile (s_memtime) of the igemm / wgrad 128x128 kernels, warm and cold
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4f; mkdir -p $O
timeout 600 python tools/kloop_profile.py 2>&1 | grep -v "amdgpu" | tee $O/kloop_warm.txt
timeout 600 python tools/kloop_profile.py cold 2>&1 | grep -v "amdgpu" | tee $O/kloop_cold.txt
