#!/bin/bash
mkdir -p gpurun_out/r6c04
O=gpurun_out/r6c04
for rep in 1 2; do for m in 0 1; do
  MPN_CONV2_CEILING=$m python tools/experiments_r6/conv2_ceiling.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events > $O/ceil${m}_$rep.json 2> $O/ceil${m}_$rep.err
  grep -o '"ms_per_step": [0-9.]*' $O/ceil${m}_$rep.json || tail -5 $O/ceil${m}_$rep.err
done; done
