#!/bin/bash
# final build: the headline step with round 6's step-level changes switched off (MPN_CONV2_CLASSES=0 = round 5's conv2) against the default,
# alternating inside one call, bench.py's own protocol (50 timed steps after 10 warm-up)
mkdir -p gpurun_out/r6c19
O=gpurun_out/r6c19
for rep in 1 2 3; do for m in 0 1; do
  MPN_CONV2_CLASSES=$m python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-events > $O/bench_cls${m}_$rep.json 2> $O/bench_cls${m}_$rep.err
  echo "conv2 classes=$m rep $rep: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' $O/bench_cls${m}_$rep.json | tr '\n' ' ')"
done; done
python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-200 $O/bench_default.json
