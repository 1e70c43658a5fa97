#!/bin/bash
# round 6 call 1: per-bucket Adam — tests, then same-call A/B of the headline step (MPN_BUCKET_ADAM 0 / 1), then force-dist
mkdir -p gpurun_out/r6c01
O=gpurun_out/r6c01
python -m pytest tests/test_round6_gpu.py tests/test_replay_gpu.py "tests/test_round2_gpu.py::test_two_rank_recorded_step_equals_the_eager_data_parallel_step" tests/test_round5_gpu.py -x -q -m gpu > $O/tests.log 2>&1
echo "tests rc $?" >> $O/tests.log
for rep in 1 2; do
  for m in 0 1; do
    MPN_BUCKET_ADAM=$m python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events > $O/bench_adam${m}_$rep.json 2> $O/bench_adam${m}_$rep.err
  done
done
python bench.py --force-dist --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events > $O/bench_force_dist.json 2> $O/bench_force_dist.err
tail -3 $O/tests.log
grep -h -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' $O/bench_*.json
