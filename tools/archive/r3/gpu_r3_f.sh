#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r3f
for V in 0 1; do PW_VARIANT=$V timeout 120 python tools/pw_timeline.py 2>&1 | grep -v amdgpu; done | tee gpurun_out/r3f/timeline.txt
