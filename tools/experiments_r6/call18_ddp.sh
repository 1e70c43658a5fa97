#!/bin/bash
mkdir -p gpurun_out/r6c18
python -m pytest tests/test_round6_gpu.py tests/test_round5_gpu.py tests/test_round2_gpu.py tests/test_round4_gpu.py -q -k "rank or dist or bucket or spawn or eager_backward" > gpurun_out/r6c18/tests.log 2>&1; echo "rc $?" >> gpurun_out/r6c18/tests.log
grep -E "passed|failed|rc " gpurun_out/r6c18/tests.log
