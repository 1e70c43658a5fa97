#!/bin/bash
# round 4, call I: epilogue load groups of 8 store iterations (GMAX 8) vs 4 — two library builds interleaved
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4i; mkdir -p $O
L=multiposenet/pytorch_amd/libmpn_hip.so
for V in old new old new old new; do
  cp tools/libmpn_$V.so $L
  timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])"
done | tee $O/step_ab.txt
for V in old new; do
  cp tools/libmpn_$V.so $L
  echo "== $V"; timeout 300 python tools/kloop_profile.py cold 2>&1 | grep -A1 "dgrad+res" | sed 's/ | span.*, / | /'
done | tee $O/kloop.txt
cp tools/libmpn_new.so $L
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_round3_gpu.py -q -x -m gpu -p no:cacheprovider -k "not trainer and not cfg5 and not cfg2" 2>&1 | tail -2
