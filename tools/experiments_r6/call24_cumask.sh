#!/bin/bash
mkdir -p gpurun_out/r6c24
O=gpurun_out/r6c24
for rep in 1 2; do for m in 0 224 192 128 64; do
  MPN_SIDE_CU_BITS=$m python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-kernel-events > $O/bench_$m_$rep.json 2> $O/bench_$m_$rep.err
  echo "side CU bits=$m rep $rep: $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$m_$rep.json || tail -2 $O/bench_$m_$rep.err)"
done; done
