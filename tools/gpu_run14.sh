#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s17; mkdir -p $O
timeout 900 python -m pytest tests/test_peaks_gpu.py tests/test_prn_assign.py tests/test_harness_gpu.py -q -x -m gpu 2>&1 | tail -5 | tee $O/tests.txt
timeout 600 python tools/infer_bench.py 2>&1 | grep -v amdgpu | tee $O/infer.txt
