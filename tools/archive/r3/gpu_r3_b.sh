#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r3b; mkdir -p $O
timeout 900 python -m pytest tests/test_round3_gpu.py tests/test_harness_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -4 $O/tests.log
timeout 300 python tools/pw_microbench.py > $O/pw_microbench.txt 2>&1; cat $O/pw_microbench.txt | grep -v amdgpu
for V in 1000000 96 1000000 96; do
  MPN_PW_MIN_TILES=$V timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pw_min_tiles=$V', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])"
done | tee $O/ab.txt
