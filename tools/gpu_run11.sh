#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s12; mkdir -p $O
for T in 64 256 1024; do
MPN_BN_FIN_MAX_TILES=$T timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events 2>&1 | tail -1 | cut -c1-260 | tee $O/bench_fin_$T.json
done
