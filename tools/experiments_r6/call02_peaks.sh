#!/bin/bash
# round 6 call 2: rewritten heat-map peak extraction — parity, microbench, kernel trace
mkdir -p gpurun_out/r6c02
O=gpurun_out/r6c02
R=$GRAFT_REPO_ROOT
python -m pytest tests/test_peaks_gpu.py tests/test_capi.py tests/test_prn_assign.py tests/test_harness_gpu.py -x -q > $O/tests.log 2>&1
echo "tests rc $?" >> $O/tests.log
python tools/peaks_microbench.py > $O/peaks_microbench.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -- python $R/tools/peaks_microbench.py > $R/$O/prof.out 2>&1
cd $R
DB=$(find $O/prof -name "*_results.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py "$DB" 1 "round 6, tools/peaks_microbench.py, rocprofv3 --kernel-trace --stats" > $O/peaks_kernel_trace.txt 2>&1
rm -rf $O/prof
tail -5 $O/tests.log; cat $O/peaks_microbench.txt; head -20 $O/peaks_kernel_trace.txt | cut -c1-200
