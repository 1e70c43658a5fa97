#!/bin/bash
mkdir -p gpurun_out/r6c05
O=gpurun_out/r6c05
python -m pytest tests/test_round6_gpu.py -x -q -k "conv2" > $O/tests_conv2.log 2>&1; echo "rc $?" >> $O/tests_conv2.log
tail -25 $O/tests_conv2.log
grep "conv2 by position" gpurun_out/parity_report.txt | tail -10
for rep in 1 2; do for m in 0 1; do
  MPN_CONV2_CLASSES=$m python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events > $O/bench_cls${m}_$rep.json 2> $O/bench_cls${m}_$rep.err
  grep -o '"ms_per_step": [0-9.]*' $O/bench_cls${m}_$rep.json || tail -5 $O/bench_cls${m}_$rep.err
done; done
