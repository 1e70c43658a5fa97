#!/bin/bash
# round 4, call B: atomic BatchNorm statistics, second form — one accumulator copy per XCD (workgroup-scope atomics) and one
# accumulator per lane (dense lines); tests + step A/B against the finalize launches
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4b; mkdir -p $O
timeout 900 python -m pytest tests/test_round4_gpu.py -q -x -m gpu -p no:cacheprovider > $O/tests_round4.log 2>&1; tail -5 $O/tests_round4.log
ab() {  # label env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])"
}
for rep in 1 2 3; do
  ab off MPN_BN_ATOMIC_STATS=0
  ab xcd8 MPN_BN_ATOMIC_STATS=1 MPN_BN_ATOMIC_XCD=1
  ab one MPN_BN_ATOMIC_STATS=1 MPN_BN_ATOMIC_XCD=0
  ab xcd8_le1000 MPN_BN_ATOMIC_STATS=1 MPN_BN_ATOMIC_XCD=1 MPN_BN_ATOMIC_MAX_TILES=1000
done 2>&1 | tee $O/bn_atomic_ab.txt
