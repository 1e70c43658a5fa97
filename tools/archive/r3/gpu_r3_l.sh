#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r3l; mkdir -p $O
timeout 900 python -m pytest tests/test_round3_gpu.py tests/test_harness_gpu.py tests/test_round2_gpu.py -x -q -m gpu -k "top_k or tester or nms" > $O/tests.log 2>&1; tail -5 $O/tests.log
timeout 300 python tools/bn_consumer_ceiling.py > $O/bn_consumer_ceiling.txt 2>&1; cat $O/bn_consumer_ceiling.txt
