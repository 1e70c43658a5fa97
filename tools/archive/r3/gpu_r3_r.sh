#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r3r; mkdir -p $O
for V in "" "mpn_bn_finalize_train,mpn_bn_bwd_finalize" "" "mpn_bn_finalize_train,mpn_bn_bwd_finalize" "mpn_bn_finalize_train" "mpn_bn_bwd_finalize"; do
  MPN_ABLATE_LAUNCHES=$V timeout 300 python tools/ablate_launches.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events 2>$O/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dropped=[$V]', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])"; tail -1 $O/err.txt
done | tee $O/ab2.txt
