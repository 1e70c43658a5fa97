#!/bin/bash
# round 4, call AB: K64 routed by rule (grid <= 512 workgroups, reduction >= 512): step A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4ab; mkdir -p $O
ab() {  # label env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])"
}
for rep in 1 2 3; do
  ab k64_0 MPN_IGEMM_K64=0
  ab k64_fwd MPN_IGEMM_K64=1
done 2>&1 | tee $O/step_ab.txt
