#!/bin/bash
# r5 call i: in-launch BatchNorm finalize with the faster last-arriver reduce (16-byte pieces, 256/(TC/2) slices): tile cap 64 (default) vs 256 vs 1024
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5i; mkdir -p $O
timeout 600 python -m pytest tests/test_round2_gpu.py tests/test_model_gpu.py -m gpu -q -x -p no:cacheprovider --timeout 500 -k "finalize or golden or train" 2>&1 | tail -2
for i in 1 2; do
  for T in 64 256 1024; do
    MPN_BN_FIN_MAX_TILES=$T timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fin_max_tiles=$T run $i', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])" | tee -a $O/ab.txt
  done
done
(cd ab_old && timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round-4 library (cap 64, old reduce)', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])") | tee -a $O/ab.txt
