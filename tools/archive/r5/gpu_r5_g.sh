#!/bin/bash
# r5 call g: cfg2 (R50 keypoint subnet 480x480 B=16 fp32) tile-selection sweep through the experiments build: is there a better tile rule for the f32 kernels?
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5g; mkdir -p $O
run() { # label, env...
  L=$1; shift
  env "$@" timeout 300 python tools/bench_experiments.py --layers 50 --size 480 --batch 16 --dtype f32 --subnet keypoint_subnet --steps 15 --warmup 4 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['ms_per_step_median_hipevent'], d['value'])" | tee -a $O/cfg2_sweep.txt
}
run default X=1
run no_tc256 MPN_TC256_MIN_BLOCKS=100000000
run tc256_from200 MPN_TC256_MIN_BLOCKS=200
run tc256_k8 MPN_TC256_MIN_KSTEPS=8
run wgrad_target256 MPN_WGRAD_TARGET=256
run wgrad_target768 MPN_WGRAD_TARGET=768
run wgrad_target1024 MPN_WGRAD_TARGET=1024
run tc128_from100 MPN_TC_MIN_BLOCKS=100
run default2 X=1
