#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r3t; mkdir -p $O
timeout 1200 python -m pytest tests/test_round3_gpu.py tests/test_round2_gpu.py tests/test_replay_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "bn_act_launch or finalize or replayed or golden or reproduc" > $O/tests.log 2>&1; tail -5 $O/tests.log
for V in 0 1 0 1 0 1; do
  MPN_BN_ACT_FINALIZE=$V timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bn_act_finalize=$V', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])"
done | tee $O/ab.txt
