#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r3i; mkdir -p $O
python tools/infer_profile2.py 2>&1 | grep -v amdgpu | tail -12
timeout 900 python tools/infer_bench.py > $O/infer_bench.txt 2>&1; grep stage $O/infer_bench.txt | cut -c1-230; grep -v stage $O/infer_bench.txt | grep -v amdgpu | tail -5
