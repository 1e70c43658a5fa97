#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r3u; mkdir -p $O
for V in "0 0" "1 0" "1 1" "1 2" "1 3" "1 4" "1 5" "1 9" "0 0"; do
  set -- $V
  MPN_BN_ACT_FINALIZE=$1 MPN_BN_ACT_FIN_DEBUG=$2 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bn_act_finalize=$1 debug=$2', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])"
done | tee $O/ab.txt
