#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r3j; mkdir -p $O
timeout 1200 python -m pytest tests/test_round2_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "finalize or fused or golden or forward" > $O/tests.log 2>&1; tail -4 $O/tests.log
for V in 64 256 64 256 1024; do
  MPN_BN_FIN_MAX_TILES=$V timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fin_max_tiles=$V', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])"
done | tee $O/ab.txt
