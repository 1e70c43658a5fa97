#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r3g; mkdir -p $O
timeout 1500 python -m pytest tests/test_round3_gpu.py tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
MPN_SIDE_STREAM=0 timeout 600 python tools/shape_report.py > $O/shape_serial_new.txt 2>&1
head -2 $O/shape_serial_new.txt | tail -1; grep -E "dgrad 1x1 256->1024|dgrad 1x1 1024->256|3x3 256->512|dgrad 3x3 256->256 @30|fwd 3x3 256->256 pyr|dgrad 3x3 256->256 pyr" $O/shape_serial_new.txt | head -20
for V in 1 2; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])"
done
