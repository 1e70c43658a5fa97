#!/bin/bash
# round 4, calls AC / AD: wave priority experiments (library built with -DMPN_MFMA_PRIO=1: over the MFMA cluster; -DMPN_DMA_PRIO=1: while queueing DMA)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4ac; mkdir -p $O
L=multiposenet/pytorch_amd/libmpn_hip.so

cp $L /tmp/base.so
ab() {  # label lib
  cp $2 $L
  timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])"
}
for rep in 1 2 3; do
  ab base /tmp/base.so
  ab prio tools/experiments_r4/libmpn_hip_prio.so
done 2>&1 | tee $O/step_ab.txt
cp tools/experiments_r4/libmpn_hip_prio.so $L
MB_ONLY=2,3,4,6 MB_COLD=1 MB_ITERS=40 timeout 300 python tools/conv_microbench.py 2>&1 | grep -v "amdgpu\|MPN_DEBUG" | tee $O/microbench_prio.txt
cp /tmp/base.so $L
MB_ONLY=2,3,4,6 MB_COLD=1 MB_ITERS=40 timeout 300 python tools/conv_microbench.py 2>&1 | grep -v "amdgpu\|MPN_DEBUG" | tee $O/microbench_base.txt
