#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s15; mkdir -p $O; : > $O/sweep.txt
MPN_IGEMM_DEEP=5 MPN_WGRAD_DEEP=5 timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -m gpu 2>&1 | tail -3 | tee $O/tests_deep5.txt
MPN_IGEMM_DEEP=4 MPN_WGRAD_DEEP=4 timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k "conv or wgrad" 2>&1 | tail -3 | tee $O/tests_deep4.txt
run() {
  local tag="$1"; shift
  local ms=$(env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step_median_hipevent'], d['ms_per_step'])")
  echo "$tag $ms" | tee -a $O/sweep.txt
}
run base X=1
run igemm4 MPN_IGEMM_DEEP=4
run igemm5 MPN_IGEMM_DEEP=5
run wgrad4 MPN_WGRAD_DEEP=4
run wgrad5 MPN_WGRAD_DEEP=5
run both5 MPN_IGEMM_DEEP=5 MPN_WGRAD_DEEP=5
run both4 MPN_IGEMM_DEEP=4 MPN_WGRAD_DEEP=4
run igemm5_768 MPN_IGEMM_DEEP=5 MPN_IGEMM_DEEP_MAX_BLOCKS=768
run base2 X=1
