#!/bin/bash
# r5 call r: every eval-mode BatchNorm's coefficients from ONE launch per pass (mpn_bn_finalize_eval_batched) instead of 104 dependent 5 us launches:
# parity (folded inference, frozen statistics, cfg5 chain), then cfg5 network A/B against the previous commit (worktree ab_old/)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5r; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider --timeout 600 -k "fold or frozen or infer or eval or cfg5 or golden or batched or tester or both" 2>&1 | tail -3
for i in 1 2; do
  for V in old new; do
    D=$GRAFT_REPO_ROOT; [ $V = old ] && D=$GRAFT_REPO_ROOT/ab_old
    (cd $D && timeout 400 python tools/infer_bench.py --iters 8 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); print('$V run $i', d['images_per_sec'], d['ms_per_batch'], d['stage'][:60])") | tee -a $O/cfg5_ab.txt
  done
done
