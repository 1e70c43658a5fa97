#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s4; mkdir -p $O
timeout 900 python -m pytest tests/test_replay_gpu.py tests/test_round2_gpu.py -m gpu -q -rf --timeout 600 -p no:cacheprovider -k "replay or graphed or adam_state or cfg3" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -v "^$" $O/pytest.log | tail -40
for L in replay eager; do
  timeout 300 python bench.py --launch $L --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events > $O/bench_$L.json 2> $O/bench_$L.err; echo "$L rc=$? $(python -c "import json;d=json.load(open('$O/bench_$L.json'));print(d['value'],d['ms_per_step'],d['ms_per_step_median_hipevent'],d['host_enqueue_ms_per_step'])")"; tail -3 $O/bench_$L.err
done
MPN_SIDE_FORK_EVERY=4 timeout 300 python bench.py --launch replay --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events > $O/bench_replay_k4.json 2> $O/bench_replay_k4.err; echo "replay k4 rc=$? $(python -c "import json;d=json.load(open('$O/bench_replay_k4.json'));print(d['value'],d['ms_per_step'],d['ms_per_step_median_hipevent'],d['host_enqueue_ms_per_step'])")"
timeout 300 python bench.py --launch replay --force-dist --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events > $O/bench_replay_dist.json 2> $O/bench_replay_dist.err; echo "replay dist rc=$? $(python -c "import json;d=json.load(open('$O/bench_replay_dist.json'));print(d['value'],d['ms_per_step'],d['ms_per_step_median_hipevent'],d['host_enqueue_ms_per_step'])")"; tail -3 $O/bench_replay_dist.err
du -sh $O
