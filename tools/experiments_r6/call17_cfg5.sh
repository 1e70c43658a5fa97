#!/bin/bash
mkdir -p gpurun_out/r6c17
for r in 1 2; do
python tools/infer_bench.py --iters 10 > gpurun_out/r6c17/cfg5_infer_bench_$r.txt 2> gpurun_out/r6c17/err_$r.txt
python -c "
import json
for l in open('gpurun_out/r6c17/cfg5_infer_bench_$r.txt'):
    d=json.loads(l); print(d['images_per_sec'], d['ms_per_batch'], d['stage'][:90])"
done
