#!/bin/bash
# r5 call j: de-phasing experiment for the loaded-epilogue igemm launches (experiments build, MPN_DEBUG_FLAGS=8192: a third of the first-round
# workgroups start 8k / 16k cycles late)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5j; mkdir -p $O
for i in 1 2 3; do
  for F in 0 8192; do
    MPN_DEBUG_FLAGS=$F timeout 300 python tools/bench_experiments.py --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('flags=$F run $i', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])" | tee -a $O/ab.txt
  done
done
