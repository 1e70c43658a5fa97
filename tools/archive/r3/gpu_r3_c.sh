#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r3c; mkdir -p $O
for V in "1000000 1" "96 1" "96 7" "1000000 1" "96 1" "96 5"; do
  set -- $V
  MPN_PW_MIN_TILES=$1 MPN_PW_EPI_MASK=$2 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pw_min_tiles=$1 mask=$2', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])"
done | tee $O/ab.txt
