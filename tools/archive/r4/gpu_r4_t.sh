#!/bin/bash
# round 4, call T: what does the weight-gradient stream cost the STEP?  recorded list without its launches (timing only: tools/ablate_launches.py)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4t; mkdir -p $O
ab() {  # label launches
  MPN_ABLATE_LAUNCHES=$2 timeout 300 python tools/ablate_launches.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])"
}
for rep in 1 2; do
  ab full ""
  ab no_wgrad mpn_conv_wgrad,mpn_conv_wgrad_partials,mpn_reduce_partials
  ab no_wgrad_no_adam mpn_conv_wgrad,mpn_conv_wgrad_partials,mpn_reduce_partials,mpn_adam_step_dev,mpn_cast_f32,mpn_weight_transpose_batched
done 2>&1 | tee $O/ablate.txt
MPN_SIDE_STREAM=0 timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('serial_schedule', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])" | tee -a $O/ablate.txt
