#!/bin/bash
# round 4, call AG: stride-2 input gradients by parity classes (MPN_DGRAD_S2_CLASSES, default 1): parity, step A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4ag; mkdir -p $O
timeout 900 python -m pytest tests/test_round4_gpu.py -q -x -m gpu -p no:cacheprovider -k "parity_class or stride2" > $O/tests_cls.log 2>&1; tail -25 $O/tests_cls.log | cut -c1-250
ab() {  # label env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])"
}
for rep in 1 2 3; do
  ab cls0 MPN_DGRAD_S2_CLASSES=0
  ab cls1 MPN_DGRAD_S2_CLASSES=1
done 2>&1 | tee $O/step_ab.txt
