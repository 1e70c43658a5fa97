#!/bin/bash
# r5 call e: same-call A/B of the whole step: commit b9cefe1 (round-4 library + NMS pin; worktree ab_old/) vs the stripped build with one fragment
# offset per tap shift in the shared-tile 3x3 kernel (HEAD); plus the conv2 / 3x3 micro-benchmark on both
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5e; mkdir -p $O
for i in 1 2 3; do
  for V in old new; do
    D=$GRAFT_REPO_ROOT; [ $V = old ] && D=$GRAFT_REPO_ROOT/ab_old
    (cd $D && timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V run $i', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])") | tee -a $O/ab.txt
  done
done
timeout 300 python -m pytest tests/test_round5_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
