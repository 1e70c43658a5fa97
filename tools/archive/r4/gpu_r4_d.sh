#!/bin/bash
# round 4, call D: atomic statistics issued BEFORE the tile's stores (mode 3) vs after (mode 1)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4d; mkdir -p $O
ab() {  # label env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])"
}
export MPN_BN_ATOMIC_XCD=0
for rep in 1 2; do
  ab off MPN_BN_ATOMIC_STATS=0
  ab m1_le1000_b2048 MPN_BN_ATOMIC_STATS=1 MPN_BN_ATOMIC_MAX_TILES=1000 MPN_BN_ACC_BLOCKS=2048
  ab m3_le1000_b2048 MPN_BN_ATOMIC_STATS=1 MPN_BN_ATOMIC_MODE=3 MPN_BN_ATOMIC_MAX_TILES=1000 MPN_BN_ACC_BLOCKS=2048
  ab m3_all_b2048 MPN_BN_ATOMIC_STATS=1 MPN_BN_ATOMIC_MODE=3 MPN_BN_ACC_BLOCKS=2048
  ab m3_le1000 MPN_BN_ATOMIC_STATS=1 MPN_BN_ATOMIC_MODE=3 MPN_BN_ATOMIC_MAX_TILES=1000
done 2>&1 | tee $O/bn_atomic_ab.txt
timeout 600 python -m pytest tests/test_round4_gpu.py -q -x -m gpu -p no:cacheprovider -k "atomic" > $O/tests.log 2>&1; tail -3 $O/tests.log
