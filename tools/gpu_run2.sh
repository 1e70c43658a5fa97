#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s2; mkdir -p $O
timeout 900 python -m pytest tests/test_round2_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -k "adam_state or two_rank or graphed" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -v "^$" $O/pytest.log | tail -60
for K in 1 4 16 64 100000; do
  MPN_GRAPH_FORK_EVERY=$K timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events > $O/bench_graph_k$K.json 2> $O/bench_graph_k$K.err; echo "graph K=$K rc=$? $(python -c "import json;d=json.load(open('$O/bench_graph_k$K.json'));print(d['value'],d['ms_per_step'],d['ms_per_step_median_hipevent'],d['host_enqueue_ms_per_step'])")"
done
for K in 4 16; do
  MPN_SIDE_FORK_EVERY=$K timeout 300 python bench.py --no-graph --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events > $O/bench_eager_k$K.json 2> $O/bench_eager_k$K.err; echo "eager K=$K rc=$? $(python -c "import json;d=json.load(open('$O/bench_eager_k$K.json'));print(d['value'],d['ms_per_step'],d['ms_per_step_median_hipevent'],d['host_enqueue_ms_per_step'])")"
done
for cfg in "--batch 32 --size 480 --mode train" "--batch 2 --size 480 --mode train" "--batch 32 --size 480 --mode eval" "--batch 8 --size 128 --mode train"; do
  timeout 300 python tools/stage_diff.py $cfg > "$O/stage_diff_$(echo $cfg | tr -d ' -').txt" 2>&1; tail -42 "$O/stage_diff_$(echo $cfg | tr -d ' -').txt"
done
du -sh $O
