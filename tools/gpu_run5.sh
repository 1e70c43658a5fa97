#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s5; mkdir -p $O
for L in replay eager graph; do
  timeout 300 python bench.py --launch $L --size 128 --batch 2 --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events > $O/small_$L.json 2> $O/small_$L.err; echo "small $L rc=$? $(python -c "import json;d=json.load(open('$O/small_$L.json'));print(d['value'],d['ms_per_step'],d['ms_per_step_median_hipevent'],d['host_enqueue_ms_per_step'])")"
done
MPN_SIDE_STREAM=0 timeout 300 python bench.py --launch replay --size 128 --batch 2 --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events > $O/small_replay_serial.json 2> $O/small_replay_serial.err; echo "small replay serial rc=$? $(python -c "import json;d=json.load(open('$O/small_replay_serial.json'));print(d['value'],d['ms_per_step'],d['ms_per_step_median_hipevent'],d['host_enqueue_ms_per_step'])")"
