#!/bin/bash
# round 4, call C: atomic statistics (single copy, one accumulator per lane) restricted by layer size; bn_act_acc grid size; TTA test
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4c; mkdir -p $O
timeout 900 python -m pytest tests/test_round4_gpu.py -q -x -m gpu -p no:cacheprovider -k "tta" > $O/tests_tta.log 2>&1; tail -5 $O/tests_tta.log
ab() {  # label env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])"
}
export MPN_BN_ATOMIC_XCD=0
for rep in 1 2; do
  ab off MPN_BN_ATOMIC_STATS=0
  ab one MPN_BN_ATOMIC_STATS=1
  ab one_le1000 MPN_BN_ATOMIC_STATS=1 MPN_BN_ATOMIC_MAX_TILES=1000
  ab one_le300 MPN_BN_ATOMIC_STATS=1 MPN_BN_ATOMIC_MAX_TILES=300
  ab one_65_1000 MPN_BN_ATOMIC_STATS=1 MPN_BN_ATOMIC_MAX_TILES=1000 MPN_BN_ATOMIC_MIN_TILES=65
  ab one_le1000_b4096 MPN_BN_ATOMIC_STATS=1 MPN_BN_ATOMIC_MAX_TILES=1000 MPN_BN_ACC_BLOCKS=4096
  ab one_le1000_b2048 MPN_BN_ATOMIC_STATS=1 MPN_BN_ATOMIC_MAX_TILES=1000 MPN_BN_ACC_BLOCKS=2048
  ab one_le1000_b1024 MPN_BN_ATOMIC_STATS=1 MPN_BN_ATOMIC_MAX_TILES=1000 MPN_BN_ACC_BLOCKS=1024
done 2>&1 | tee $O/bn_atomic_ab.txt
