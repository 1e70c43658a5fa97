#!/bin/bash
mkdir -p gpurun_out/r6c12
O=gpurun_out/r6c12
python -m pytest tests/test_round6_gpu.py -x -q -k "training_curves" > $O/test_train.log 2>&1; echo "rc $?" >> $O/test_train.log
grep -E "passed|failed|rc |assert" $O/test_train.log | head; grep "does it train" gpurun_out/parity_report.txt | tail -2
python tools/infer_bench.py --iters 10 > $O/cfg5_infer_bench.txt 2> $O/cfg5_infer_bench.err
cut -c1-200 $O/cfg5_infer_bench.txt
python bench.py --layers 101 --size 800 --batch 8 --steps 15 --warmup 4 --no-cpu-baseline > $O/cfg4_bench.json 2> $O/cfg4_bench.err
cut -c1-400 $O/cfg4_bench.json
