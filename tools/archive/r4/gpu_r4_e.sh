#!/bin/bash
# round 4, call E: are the k-loops bound by HBM latency?  warm (operands MALL-resident) vs cold micro-benchmarks; atomic-BN defaults A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4e; mkdir -p $O
for cold in 0 1; do
  echo "== MB_COLD=$cold"; MB_ONLY=2,3,4,5 MB_COLD=$cold MB_ITERS=40 timeout 300 python tools/conv_microbench.py 2>&1 | grep -v "amdgpu\|DEBUG"
done | tee $O/microbench_warm_cold.txt
ab() {  # label env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])"
}
for rep in 1 2 3; do
  ab off MPN_BN_ATOMIC_STATS=0
  ab default
done 2>&1 | tee $O/bn_atomic_default_ab.txt
timeout 900 python -m pytest tests/test_round4_gpu.py -q -x -m gpu -p no:cacheprovider > $O/tests.log 2>&1; tail -3 $O/tests.log
