#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r3n; mkdir -p $O
for V in unset 0 unset 0; do
  if [ $V = unset ]; then unset MPN_MAIN_PRIORITY; else export MPN_MAIN_PRIORITY=$V; fi
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('main stream: $V', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])"
done | tee -a $O/ab.txt
