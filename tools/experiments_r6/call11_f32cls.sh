#!/bin/bash
mkdir -p gpurun_out/r6c11
O=gpurun_out/r6c11
python -m pytest tests/test_round6_gpu.py -x -q -k "conv2" > $O/tests_conv2.log 2>&1; echo "rc $?" >> $O/tests_conv2.log
grep -E "passed|failed|rc |Error|error" $O/tests_conv2.log | head; grep "fp32," gpurun_out/parity_report.txt | tail -3
CFG2="--layers 50 --size 480 --batch 16 --dtype f32 --subnet keypoint_subnet"
for rep in 1 2; do for m in 0 1; do
  MPN_CONV2_CLASSES=$m python bench.py $CFG2 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events > $O/cfg2_cls${m}_$rep.json 2> $O/cfg2_cls${m}_$rep.err
  echo "cfg2 classes=$m: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' $O/cfg2_cls${m}_$rep.json | tr '\n' ' ' || tail -3 $O/cfg2_cls${m}_$rep.err)"
done; done
for m in 0 1; do
  MPN_CONV2_FWD_TAPS=$m python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events > $O/bench_taps${m}.json 2> $O/bench_taps${m}.err
  echo "headline fwd_taps=$m: $(grep -o '"ms_per_step": [0-9.]*' $O/bench_taps${m}.json || tail -3 $O/bench_taps${m}.err)"
done
python tools/experiments_r6/one_by_one_tail_ceiling.py > $O/one_by_one_tail_ceiling.txt 2>&1; tail -12 $O/one_by_one_tail_ceiling.txt
