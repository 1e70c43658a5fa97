#!/bin/bash
# in-call A/B: library with the round-2-style one-level fin_last_arriver (conv_igemm.hip of 600c0bc) vs the final one
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
L=multiposenet/pytorch_amd/libmpn_hip.so
for V in old new old new old new; do
  cp tools/libmpn_$V.so $L
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])"
done | tee gpurun_out/r3x_ab.txt
cp tools/libmpn_new.so $L
