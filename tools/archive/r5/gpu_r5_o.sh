#!/bin/bash
# r5 call o: re-sweep of the tuned constants at step level through the experiments build (the kernels changed since the round-2 / round-3 sweeps)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5o; mkdir -p $O
run() { L=$1; shift
  env "$@" timeout 300 python tools/bench_experiments.py --steps 25 --warmup 8 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-28s' % '$L', d['ms_per_step_median_hipevent'], d['value'])" | tee -a $O/sweep.txt
}
run default X=1
run tc_min_blocks=150 MPN_TC_MIN_BLOCKS=150
run tc_min_blocks=300 MPN_TC_MIN_BLOCKS=300
run tc256_min_blocks=300 MPN_TC256_MIN_BLOCKS=300
run tc256_min_blocks=600 MPN_TC256_MIN_BLOCKS=600
run tc256_min_ksteps=8 MPN_TC256_MIN_KSTEPS=8
run tc256_min_ksteps=32 MPN_TC256_MIN_KSTEPS=32
run default X=1
run wgrad_target=448 MPN_WGRAD_TARGET=448
run wgrad_target=640 MPN_WGRAD_TARGET=640
run wgrad_minpix=256 MPN_WGRAD_MINPIX=256
run wgrad_minpix=1024 MPN_WGRAD_MINPIX=1024
run bn_blocks=4096 MPN_BN_BLOCKS=4096
run bn_blocks=16384 MPN_BN_BLOCKS=16384
run default X=1
