#!/bin/bash
# round 6 call 3: cfg5 end to end with the rewritten peak extraction + kernel trace
mkdir -p gpurun_out/r6c03
O=gpurun_out/r6c03
R=$GRAFT_REPO_ROOT
python tools/infer_bench.py --iters 10 > $O/cfg5_infer_bench.txt 2> $O/cfg5_infer_bench.err
cat $O/cfg5_infer_bench.txt | cut -c1-260
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof -- python $R/tools/infer_bench.py --iters 4 > $R/$O/prof.out 2>&1
cd $R
DB=$(find $O/prof -name "*_results.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py "$DB" 25 "round 6, cfg5: python tools/infer_bench.py --iters 4 (R101 both 640x640 B=64 f16, BN folded; 5 stage sets x (1 + 4) batches = 25 batches; calibration passes included), rocprofv3 --kernel-trace --stats" > $O/cfg5_kernel_trace.txt 2>&1
rm -rf $O/prof
head -50 $O/cfg5_kernel_trace.txt | cut -c1-180
