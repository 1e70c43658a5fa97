#!/bin/bash
# round 4, call O: LDS-DMA without the M0 save / restore (new) vs with (old), two library builds interleaved; wgrad slice target 500
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4o; mkdir -p $O
L=multiposenet/pytorch_amd/libmpn_hip.so
run() { timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])"; }
for V in old new old new old new; do cp tools/libmpn_$V.so $L; run $V; done | tee $O/step_ab.txt
cp tools/libmpn_new.so $L
for T in 512 500 448; do MPN_WGRAD_TARGET=$T run target$T; done | tee -a $O/step_ab.txt
for V in old new; do cp tools/libmpn_$V.so $L; echo "== $V"; timeout 300 python tools/kloop_profile.py 2>&1 | grep -A3 "^3x3 256->256 @30\|^1x1 1024" | sed 's/ | span.*, / | /' | grep -v production; done | tee $O/kloop.txt
cp tools/libmpn_new.so $L
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q -x -m gpu -p no:cacheprovider 2>&1 | tail -2
