#!/bin/bash
# round 4, call J: epilogue operand touch (MPN_EPI_TOUCH=1) vs off: parity, the loaded dgrad's phase profile, step A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4j; mkdir -p $O
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_round3_gpu.py -q -x -m gpu -p no:cacheprovider -k "not trainer and not cfg5 and not cfg2" 2>&1 | tail -2
for V in 0 1; do
  echo "== MPN_EPI_TOUCH=$V"; MPN_EPI_TOUCH=$V timeout 300 python tools/kloop_profile.py cold 2>&1 | grep -A1 "dgrad" | grep -v "^--" | sed 's/ | span.*, / | /'
done | tee $O/kloop.txt
ab() {  # label env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])"
}
for rep in 1 2 3; do
  ab touch0 MPN_EPI_TOUCH=0
  ab touch1 MPN_EPI_TOUCH=1
done 2>&1 | tee $O/step_ab.txt
