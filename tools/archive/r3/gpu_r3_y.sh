#!/bin/bash
# knob sweep at the step level on the final build (defaults first and last)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r3y; mkdir -p $O
run() { env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])"; }
{
run X=0
run MPN_WGRAD_TARGET=384
run MPN_WGRAD_TARGET=768
run MPN_WGRAD_MINPIX=1024
run MPN_SIDE_FORK_EVERY=2
run MPN_SIDE_FORK_EVERY=4
run MPN_TC_MIN_BLOCKS=300
run MPN_TC_MIN_BLOCKS=128
run MPN_WGRAD_TM256_MIN_STEPS=64
run X=0
} | tee $O/sweep.txt
