#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r3m; mkdir -p $O
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())" 2>&1 | tail -1 | tee $O/ab.txt
timeout 600 python -m pytest tests/test_harness_gpu.py -x -q -m gpu -k "tester" 2>&1 | tail -2
for V in "0 0" "-1 0" "0 -1" "0 0" "-1 0" "0 -1"; do
  set -- $V
  MPN_MAIN_PRIORITY=$1 MPN_SIDE_PRIORITY=$2 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('main_prio=$1 side_prio=$2', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])"
done | tee -a $O/ab.txt
