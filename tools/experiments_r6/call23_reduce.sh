#!/bin/bash
mkdir -p gpurun_out/r6c23
O=gpurun_out/r6c23
python -m pytest tests/test_kernels_gpu.py tests/test_round4_gpu.py tests/test_replay_gpu.py tests/test_model_gpu.py -q -x -k "wgrad or replayed or gradient or golden or train" > $O/tests.log 2>&1; echo "rc $?" >> $O/tests.log
grep -E "passed|failed|rc " $O/tests.log
make -s -j8 -C multiposenet/pytorch_amd/csrc experiments > $O/make_exp.log 2>&1
for rep in 1 2 3; do for m in 0 -1; do
  MPN_REDUCE_SL=$m python tools/bench_experiments.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-events > $O/bench_sl${m}_$rep.json 2> $O/bench_sl${m}_$rep.err
  echo "reduce slices=$m rep $rep: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' $O/bench_sl${m}_$rep.json | tr '\n' ' ')"
done; done
