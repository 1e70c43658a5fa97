#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r3h; mkdir -p $O
for V in 16 8 16 8; do
  MPN_TC256_MIN_KSTEPS=$V timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tc256_min_ksteps=$V', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])"
done | tee $O/ab.txt
