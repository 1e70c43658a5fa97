#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r3p67; mkdir -p $O
for V in 1 2 1 2 1 2; do
  MPN_DET_PYRAMID_SIDE=$V timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('p67_side=$V', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])"
done | tee $O/ab_whole.txt
