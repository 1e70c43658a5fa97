#!/bin/bash
mkdir -p gpurun_out/r6c22
python -m pytest tests -q -m gpu -k "bench or force_dist or spawn or rendezvous" > gpurun_out/r6c22/tests.log 2>&1; echo "rc $?" >> gpurun_out/r6c22/tests.log
grep -E "passed|failed|rc " gpurun_out/r6c22/tests.log
python bench.py --steps 5 --warmup 2 --layers 50 --size 256 --batch 4 --no-cpu-baseline --no-kernel-events | cut -c1-700
