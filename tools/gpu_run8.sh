#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s9; mkdir -p $O
for cfg in "0 1" "20 1" "40 1" "40 2" "60 1" "80 1"; do
  set -- $cfg
  MPN_SIDE_DEFER=$1 MPN_SIDE_RELEASE=$2 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events > $O/b_$1_$2.json 2> $O/b_$1_$2.err; echo "defer=$1 release=$2 rc=$? $(python -c "import json;d=json.load(open('$O/b_$1_$2.json'));print(d['value'],d['ms_per_step'],d['ms_per_step_median_hipevent'])")"
done
