#!/bin/bash
mkdir -p gpurun_out/r6c10
O=gpurun_out/r6c10
python -m pytest tests/test_round6_gpu.py -x -q -k "conv2" > $O/tests_conv2.log 2>&1; echo "rc $?" >> $O/tests_conv2.log
MPN_CONV2_CLS_STREAM=1 python -m pytest tests/test_round6_gpu.py tests/test_replay_gpu.py -x -q -k "conv2 or replayed" >> $O/tests_conv2.log 2>&1; echo "rc $?" >> $O/tests_conv2.log
grep -E "passed|failed|rc " $O/tests_conv2.log
for rep in 1 2; do for m in "0 0" "1 0" "1 1"; do
  set -- $m
  MPN_CONV2_CLASSES=$1 MPN_CONV2_CLS_STREAM=$2 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events > $O/bench_$1$2_$rep.json 2> $O/bench_$1$2_$rep.err
  echo "classes=$1 own_stream=$2: $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$1$2_$rep.json || tail -3 $O/bench_$1$2_$rep.err)"
done; done
