#!/bin/bash
mkdir -p gpurun_out/r6c08
python -m pytest tests/ -x -q -m gpu > gpurun_out/r6c08/gpu_tests.log 2>&1; echo "rc $?" >> gpurun_out/r6c08/gpu_tests.log
tail -15 gpurun_out/r6c08/gpu_tests.log
