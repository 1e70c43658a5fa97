#!/bin/bash
mkdir -p gpurun_out/r6c15
O=gpurun_out/r6c15
python -m pytest tests/test_round6_gpu.py tests/test_round3_gpu.py -q -k "pipelined or cfg5_chain or bbox_transform" > $O/tests.log 2>&1; echo "rc $?" >> $O/tests.log
grep -E "passed|failed|rc |Error" $O/tests.log | head -8
python tools/experiments_r6/pipeline_timeline.py 2>&1 | tail -2
