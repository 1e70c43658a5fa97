#!/bin/bash
# round 4, call S: wgrad slice count rounded DOWN to at most 512 workgroups (new default) vs up (MPN_WGRAD_CEIL=1)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4s; mkdir -p $O
ab() {  # label env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])"
}
for rep in 1 2 3 4; do
  ab ceil MPN_WGRAD_CEIL=1
  ab floor MPN_WGRAD_CEIL=0
done 2>&1 | tee $O/step_ab.txt
for v in 1 0; do echo "== microbench MPN_WGRAD_CEIL=$v"; MPN_WGRAD_CEIL=$v MB_COLD=1 MB_ITERS=40 timeout 300 python tools/conv_microbench.py 2>&1 | grep -v "amdgpu\|DEBUG"; done | tee $O/microbench.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_replay_gpu.py -q -x -m gpu -p no:cacheprovider 2>&1 | tail -2
