#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6c21; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-events > $R/$O/prof.out 2>&1
cd $R
DB=$(find $O/prof -name "*_results.db" | head -1)
python tools/prof_busy.py "$DB" > $O/gpu_busy.txt 2>&1
rm -rf $O/prof
cat $O/gpu_busy.txt
