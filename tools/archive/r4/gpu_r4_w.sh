#!/bin/bash
# round 4, call W: the operand path's delivery rate (tools/microbench/dma_rate) and the weight-gradient slice target at 3 workgroups per CU
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4w; mkdir -p $O
timeout 300 tools/microbench/dma_rate 40 2>&1 | tee $O/dma_rate_40k.txt
timeout 300 tools/microbench/dma_rate 8 2>&1 | tee $O/dma_rate_8k.txt
timeout 300 tools/microbench/dma_rate 512 2>&1 | tee $O/dma_rate_512k.txt
for v in 512 768 512 768; do
  echo "== microbench MPN_WGRAD_TARGET=$v"; MPN_WGRAD_TARGET=$v MB_ONLY=2,3,4,5,6 MB_COLD=1 MB_ITERS=40 timeout 300 python tools/conv_microbench.py 2>&1 | grep "wgrad"
done | tee $O/microbench_target.txt
ab() {  # label env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])"
}
for rep in 1 2; do
  ab t512 MPN_WGRAD_TARGET=512
  ab t768 MPN_WGRAD_TARGET=768
done 2>&1 | tee $O/step_ab.txt
