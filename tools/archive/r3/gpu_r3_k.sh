#!/bin/bash
# one-pass heat-map loss for the recorded step: parity tests, then A/B of the step with and without it inside one call
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r3k; mkdir -p $O
timeout 1500 python -m pytest tests/test_round3_gpu.py tests/test_replay_gpu.py tests/test_round2_gpu.py -x -q -m gpu -k "one_pass or replay or recorded or trainer_uses" > $O/tests.log 2>&1; tail -6 $O/tests.log
for V in 0 1 0 1 0 1; do
  MPN_FUSED_MSE=$V timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused_mse=$V', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])"
done | tee $O/ab.txt
