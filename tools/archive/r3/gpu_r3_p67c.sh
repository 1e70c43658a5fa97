#!/bin/bash
# whole detection pyramid on the side stream in the forward pass (MPN_DET_PYRAMID_SIDE=2): parity suites, then a second A/B against P6/P7 only
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r3p67; mkdir -p $O
MPN_DET_PYRAMID_SIDE=2 timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_replay_gpu.py tests/test_round2_gpu.py -x -q -m gpu -p no:cacheprovider -k "golden or replayed or reproduc or cfg5 or overlap or serial or two_rank" > $O/tests_whole.log 2>&1; tail -2 $O/tests_whole.log
for V in 1 2 1 2 1 2 1 2; do
  MPN_DET_PYRAMID_SIDE=$V timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('p67_side=$V', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])"
done | tee $O/ab_whole2.txt
