#!/bin/bash
# HBM traffic per kernel: rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in SEPARATE passes (MI355X_MICROARCH.md, HBM section) over the
# bench command; summary -> gpurun_out/$1/pmc_hbm_traffic.{json,txt}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
TAG=${1:-pmc}; O=gpurun_out/$TAG; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  MPN_SIDE_STREAM=0 timeout 900 rocprofv3 --kernel-trace --pmc $C -d $R/$O/$C -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events > $R/$O/$C.out 2>&1
  grep '"metric"' $R/$O/$C.out | cut -c1-200
done
cd $R
F=$(find $O/FETCH_SIZE -name "*_results.db" | head -1); W=$(find $O/WRITE_SIZE -name "*_results.db" | head -1)
# bench: 2 set-up + 1 warm-up + 3 timed + 5 empty-queue host measurements = 11 steps
python tools/pmc_summary.py "$F" "$W" 11 $O/pmc_hbm_traffic.json > $O/pmc_hbm_traffic.txt 2>&1
python - <<PY
import json, sys
sys.path.insert(0, "$R")
from multiposenet.pytorch_amd import _lib
p = "$O/pmc_hbm_traffic.json"
d = json.load(open(p)); d["build_id"] = _lib.lib().mpn_version().decode()
json.dump(d, open(p, "w"), indent=1)
PY
rm -rf $O/FETCH_SIZE $O/WRITE_SIZE
head -25 $O/pmc_hbm_traffic.txt
