#!/bin/bash
# one bench line per tuning-knob setting (A/B against "base" in the same call)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/sweep; mkdir -p $O; : > $O/sweep.txt
run() {
  local tag="$1"; shift
  local ms=$(env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step_median_hipevent'], d['ms_per_step'])")
  echo "$tag $ms" | tee -a $O/sweep.txt
}
run base X=1
run wgrad_target_384 MPN_WGRAD_TARGET=384
run wgrad_target_768 MPN_WGRAD_TARGET=768
run wgrad_target_1024 MPN_WGRAD_TARGET=1024
run wgrad_minpix_1024 MPN_WGRAD_MINPIX=1024
run tc256_ksteps_8 MPN_TC256_MIN_KSTEPS=8
run tc256_blocks_256 MPN_TC256_MIN_BLOCKS=256
run tc256_blocks_800 MPN_TC256_MIN_BLOCKS=800
run tc_min_blocks_400 MPN_TC_MIN_BLOCKS=400
run tc_min_blocks_100 MPN_TC_MIN_BLOCKS=100
run fork_every_2 MPN_SIDE_FORK_EVERY=2
run fork_every_4 MPN_SIDE_FORK_EVERY=4
run base2 X=1
