#!/bin/bash
# round 4, call AI: weight-gradient slice target below 2 workgroups per CU (fewer partial sums, more room for the main stream)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4ai; mkdir -p $O
ab() {  # label env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])"
}
for rep in 1 2; do
  ab t512 MPN_WGRAD_TARGET=512
  ab t384 MPN_WGRAD_TARGET=384
  ab t256 MPN_WGRAD_TARGET=256
done 2>&1 | tee $O/step_ab.txt
