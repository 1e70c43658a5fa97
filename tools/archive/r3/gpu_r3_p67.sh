#!/bin/bash
# P6 / P7 of the detection pyramid on the side stream in the forward pass: parity suites, then step A/B (first call: MPN_DET_PYRAMID_SIDE;
# second call, as committed: the coarse keypoint-head branches, MPN_KP_COARSE_SIDE — neutral, removed again)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r3p67; mkdir -p $O
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_replay_gpu.py tests/test_round2_gpu.py -x -q -m gpu -p no:cacheprovider -k "golden or replayed or reproduc or cfg5 or overlap or serial or two_rank" > $O/tests_kp.log 2>&1; tail -4 $O/tests_kp.log
for V in 0 1 0 1 0 1; do
  MPN_KP_COARSE_SIDE=$V timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('kp_coarse_side=$V', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])"
done | tee $O/ab_kp.txt
