#!/bin/bash
# ceiling of any in-kernel split-K reduction: the step with (a) the reduction launches removed, (b) the partial stores removed as well
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r3e; mkdir -p $O
for V in 0 1 3 0 1 3; do
  MPN_WGRAD_ABLATE=$V timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wgrad_ablate=$V', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])"
done | tee $O/ab.txt
for V in 0 1 3; do
  MPN_SIDE_STREAM=0 MPN_WGRAD_ABLATE=$V timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('serial schedule wgrad_ablate=$V', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])"
done | tee -a $O/ab.txt
