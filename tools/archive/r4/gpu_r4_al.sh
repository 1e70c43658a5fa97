#!/bin/bash
# round 4, call AL: ceilings by launch ablation on the final build — the five separate BatchNorm-backward reductions, the ten ReLU-backward launches, the 2 x 95 finalize launches
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4al; mkdir -p $O
ab() {  # label launches
  MPN_ABLATE_LAUNCHES=$2 timeout 300 python tools/ablate_launches.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step_median_hipevent'], d['ms_per_step'])"
}
for rep in 1 2; do
  ab full ""
  ab no_bn_bwd_reduce mpn_bn_bwd_reduce
  ab no_relu_backward mpn_relu_backward
  ab no_finalize mpn_bn_finalize_train,mpn_bn_bwd_finalize
  ab no_upsample_bwd mpn_upsample_nearest_backward,mpn_upsample_nearest_slice_backward
done 2>&1 | tee $O/ablation.txt
