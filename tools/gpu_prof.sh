#!/bin/bash
# kernel traces (serial + overlapped) and the per-shape conv report of the current build -> gpurun_out/$1
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
TAG=${1:-prof}; O=gpurun_out/$TAG; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp
MPN_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/serial -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-events > $R/$O/serial.out 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/overlap -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-events > $R/$O/overlap.out 2>&1
cd $R
for m in serial overlap; do
  DB=$(find $O/$m -name "*_results.db" | head -1)
  # bench: 2 set-up (1 eager + 1 recording) + 4 warm-up + 12 timed + 5 empty-queue host measurements = 23 steps
  [ -n "$DB" ] && python tools/rocprof_summary.py "$DB" 23 "round ${MPN_ROUND:-6}, $m schedule, python bench.py --steps 12 --warmup 4 (23 steps incl. set-up and the 5 empty-queue host measurements), rocprofv3 --kernel-trace --stats" > $O/kernel_trace_$m.txt 2>&1
  [ -n "$DB" ] && rm -rf $O/$m
  grep '"metric"' $O/$m.out | cut -c1-300
done
head -45 $O/kernel_trace_serial.txt | cut -c1-190
timeout 600 python tools/shape_report.py > $O/shape_report.txt 2>&1; head -70 $O/shape_report.txt
du -sh $O
