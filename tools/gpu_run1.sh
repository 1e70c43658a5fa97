#!/bin/bash
# round-2 GPU session 1: new tests, graph vs eager bench, RCCL leg, kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s1; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -25 $O/pytest.log
timeout 600 python bench.py --steps 50 --warmup 10 > $O/bench_graph.json 2> $O/bench_graph.err; echo "graph rc=$?"; cat $O/bench_graph.json | cut -c1-600
timeout 300 python bench.py --no-graph --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events > $O/bench_eager.json 2> $O/bench_eager.err; echo "eager rc=$?"; cut -c1-400 $O/bench_eager.json
MPN_SIDE_STREAM=0 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events > $O/bench_graph_serial.json 2> $O/bench_graph_serial.err; echo "graph-serial rc=$?"; cut -c1-400 $O/bench_graph_serial.json
timeout 300 python bench.py --force-dist --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events > $O/bench_dist.json 2> $O/bench_dist.err; echo "dist rc=$?"; cut -c1-400 $O/bench_dist.json; tail -5 $O/bench_dist.err
timeout 300 python bench.py --force-dist --no-graph --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events > $O/bench_dist_eager.json 2> $O/bench_dist_eager.err; echo "dist-eager rc=$?"; cut -c1-400 $O/bench_dist_eager.json
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_graph -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-events > $GRAFT_REPO_ROOT/$O/prof_graph.out 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT; ls $O/prof_graph | head; du -sh $O
DB=$(find $O/prof_graph -name "*_results.db" | head -1); echo "db=$DB"
[ -n "$DB" ] && python tools/rocprof_summary.py "$DB" 16 "round 2, graph replay bench (4 set-up + 4 warm-up... see prof_graph.out), rocprofv3 --kernel-trace --stats" > $O/prof_graph_summary.txt 2>&1 && head -30 $O/prof_graph_summary.txt | cut -c1-200
[ -n "$DB" ] && rm -f "$DB"
