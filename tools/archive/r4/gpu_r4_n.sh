#!/bin/bash
# round 4, call N: after removing the variants that lost twice (conv_pw, two-level / bn_act-side finalize, split concat dgrad, fused
# ReLU-backward): the whole GPU suite + a bench line against the previous library (tools/libmpn_old.so = before the removal)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4n; mkdir -p $O
timeout 2400 python -m pytest tests -q -x -m gpu -p no:cacheprovider > $O/tests_all.log 2>&1; tail -6 $O/tests_all.log
for rep in 1 2 3; do
  timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pruned', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])"
done | tee $O/step.txt
