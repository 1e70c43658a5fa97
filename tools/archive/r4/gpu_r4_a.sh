#!/bin/bash
# round 4, call A: atomic BatchNorm statistics — new parity tests, the existing model gates on the new default, step A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4a; mkdir -p $O
timeout 900 python -m pytest tests/test_round4_gpu.py -q -x -m gpu -p no:cacheprovider > $O/tests_round4.log 2>&1; tail -5 $O/tests_round4.log
ab() {  # label env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])"
}
for rep in 1 2 3; do
  ab off MPN_BN_ATOMIC_STATS=0
  ab on MPN_BN_ATOMIC_STATS=1
  ab on_gt64 MPN_BN_ATOMIC_STATS=1 MPN_BN_ATOMIC_MIN_TILES=65
  ab on_le1000 MPN_BN_ATOMIC_STATS=1 MPN_BN_ATOMIC_MAX_TILES=1000
done 2>&1 | tee $O/bn_atomic_ab.txt
ab on_blocks4096 MPN_BN_ATOMIC_STATS=1 MPN_BN_ACC_BLOCKS=4096 | tee -a $O/bn_atomic_ab.txt
ab on_blocks2048 MPN_BN_ATOMIC_STATS=1 MPN_BN_ACC_BLOCKS=2048 | tee -a $O/bn_atomic_ab.txt
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_replay_gpu.py -q -x -m gpu -p no:cacheprovider > $O/tests_model.log 2>&1; tail -5 $O/tests_model.log
