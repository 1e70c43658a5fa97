#!/bin/bash
# r5 call n: f32 ring weight gradient with all fragment reads of a k-step issued first + scalar bias branch (both dtypes): parity, then same-call A/B
# of cfg2 and of the headline against the previous commit (worktree ab_old/)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5n; mkdir -p $O
timeout 900 python -m pytest tests/test_round4_gpu.py tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -x -p no:cacheprovider --timeout 600 -k "wgrad or golden or train or step or grad" 2>&1 | tail -2
for i in 1 2; do
  for V in old new; do
    D=$GRAFT_REPO_ROOT; [ $V = old ] && D=$GRAFT_REPO_ROOT/ab_old
    (cd $D && timeout 300 python bench.py --layers 50 --size 480 --batch 16 --dtype f32 --subnet keypoint_subnet --steps 15 --warmup 4 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg2 $V run $i', d['ms_per_step_median_hipevent'], d['value'])") | tee -a $O/ab.txt
  done
done
for i in 1 2 3; do
  for V in old new; do
    D=$GRAFT_REPO_ROOT; [ $V = old ] && D=$GRAFT_REPO_ROOT/ab_old
    (cd $D && timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline $V run $i', d['ms_per_step_median_hipevent'], d['value'])") | tee -a $O/ab.txt
  done
done
