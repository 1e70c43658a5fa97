#!/bin/bash
# round 4, call P: does the k-step time of a wave depend on how many workgroups share the CU?  wgrad slice target 256 / 512 / 768 / 1024
# workgroups (1 / 2 / 3 / 3+ per CU) on the same shapes, phase profile + duration
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4p; mkdir -p $O
for T in 256 512 768 1024; do
  echo "== MPN_WGRAD_TARGET=$T MPN_WGRAD_MINPIX=64"
  MPN_WGRAD_TARGET=$T MPN_WGRAD_MINPIX=64 timeout 300 python tools/kloop_profile.py 2>&1 | grep "wgrad\|^[13]x" | sed 's/ | span.*, / | /'
done | tee $O/wgrad_occupancy.txt
