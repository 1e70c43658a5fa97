#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4l; mkdir -p $O
timeout 900 python -m pytest tests/test_round4_gpu.py -q -x -m gpu -p no:cacheprovider -k "pyramids_heads" > $O/tests.log 2>&1; tail -30 $O/tests.log; grep "(i\|bf16 pyramids" gpurun_out/parity_report.txt | tail -8
