#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s13; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_round2_gpu.py -q -x -m gpu -k "conv or bn or folded or pyramid or f16" 2>&1 | tail -5 | tee $O/tests.txt
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events 2>&1 | tail -1 | cut -c1-400 | tee $O/bench.json
timeout 600 python tools/shape_report.py > $O/shape_report.txt 2>&1; head -40 $O/shape_report.txt
