#!/bin/bash
mkdir -p gpurun_out/r6c06
O=gpurun_out/r6c06
R=$GRAFT_REPO_ROOT
python -m pytest tests/test_round6_gpu.py -x -q -k "conv2" > $O/tests_conv2.log 2>&1; echo "rc $?" >> $O/tests_conv2.log
tail -3 $O/tests_conv2.log
MPN_SIDE_STREAM=0 python tools/shape_report.py > $O/shape_cls.txt 2>&1
grep -n "virtual-cat\|2304\|@120x120\|total" $O/shape_cls.txt | head -40
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-events > $R/$O/prof.out 2>&1
cd $R
DB=$(find $O/prof -name "*_results.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py "$DB" 23 "round 6 conv2 classes, overlap schedule" > $O/kernel_trace_overlap.txt 2>&1
rm -rf $O/prof
grep -n "conv2cls\|upsample_slice\|fill_f32\|cast_f32\|weight_transpose_kernel\|relu_bwd" $O/kernel_trace_overlap.txt | cut -c1-180
