#!/bin/bash
# round-end artefacts: default bench line (with cpu_baseline), 1-rank RCCL line, serial/overlap kernel traces, serial shape report
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
TAG=${1:-final}; O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-1500
MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 timeout 400 python bench.py --force-dist --no-cpu-baseline --no-kernel-events > $O/bench_dist.json 2> $O/bench_dist.err; tail -1 $O/bench_dist.json | cut -c1-200
bash tools/gpu_prof.sh $TAG > $O/prof.log 2>&1
MPN_SIDE_STREAM=0 timeout 600 python tools/shape_report.py > $O/shape_report_serial.txt 2>&1
head -30 $O/kernel_trace_serial.txt | cut -c1-170
