#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r3d; mkdir -p $O; rm -f gpurun_out/parity_report.txt
timeout 1200 python -m pytest tests/test_round3_gpu.py -q -m gpu -k "bf16" > $O/tests.log 2>&1; tail -5 $O/tests.log
cat gpurun_out/parity_report.txt | tail -22
