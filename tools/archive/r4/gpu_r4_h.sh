#!/bin/bash
# round 4, call H: phase profile incl. the conv1 input-gradient launch with its loaded epilogue (res through mask bits + BN-backward statistics)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4h; mkdir -p $O
timeout 600 python tools/kloop_profile.py cold 2>&1 | grep -v "amdgpu" | sed 's/ | span.*, / | /' | tee $O/kloop_cold.txt | head -24
