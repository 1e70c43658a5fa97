#!/bin/bash
# the same bench at BASELINE.json's other training shapes (one line each) + the config-5 inference chain
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r3cfg; mkdir -p $O
run() { timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', '->', d['value'], 'img/s', d['ms_per_step'], 'ms/step')"; }
{
run --layers 101 --size 480 --batch 32 --dtype bf16
run --layers 50 --size 480 --batch 32 --dtype bf16
run --layers 50 --size 480 --batch 16 --dtype f32
run --layers 101 --size 480 --batch 32 --dtype f32
run --layers 101 --size 800 --batch 8 --dtype bf16
run --layers 101 --size 480 --batch 32 --dtype bf16 --eager-log
run --layers 101 --size 480 --batch 32 --dtype bf16 --launch eager
} | tee $O/configs.txt
timeout 600 python tools/infer_bench.py --iters 10 2>/dev/null | tee $O/infer.txt
