#!/bin/bash
# serial kernel trace + PMC HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes) + per-kernel bandwidth table of ONE bench
# configuration:  tools/gpu_profile_config.sh TAG "bench flags"   ->  gpurun_out/TAG/{kernel_trace_serial,pmc_hbm_traffic,hbm_bandwidth_per_kernel}.txt
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
TAG=$1; FLAGS=$2; O=gpurun_out/$TAG; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp
MPN_SIDE_STREAM=0 timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/serial -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-events $FLAGS > $R/$O/serial.out 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  MPN_SIDE_STREAM=0 timeout 900 rocprofv3 --kernel-trace --pmc $C -d $R/$O/$C -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events $FLAGS > $R/$O/$C.out 2>&1
done
cd $R
DB=$(find $O/serial -name "*_results.db" | head -1)
# bench: 2 set-up (1 eager + 1 recording) + 2 warm-up + 6 timed + 5 empty-queue host measurements = 15 steps
python tools/rocprof_summary.py "$DB" 15 "round ${MPN_ROUND:-6}, serial schedule (MPN_SIDE_STREAM=0), python bench.py --steps 6 --warmup 2 $FLAGS (15 steps incl. set-up and the host measurements), rocprofv3 --kernel-trace --stats" > $O/kernel_trace_serial.txt 2>&1
F=$(find $O/FETCH_SIZE -name "*_results.db" | head -1); W=$(find $O/WRITE_SIZE -name "*_results.db" | head -1)
python tools/pmc_summary.py "$F" "$W" 11 $O/pmc_hbm_traffic.json > $O/pmc_hbm_traffic.txt 2>&1
python - <<PY
import json, sys
sys.path.insert(0, "$R")
from multiposenet.pytorch_amd import _lib
p = "$O/pmc_hbm_traffic.json"
d = json.load(open(p)); d["build_id"] = _lib.lib().mpn_version().decode(); d["bench_flags"] = "$FLAGS"
json.dump(d, open(p, "w"), indent=1)
PY
python tools/hbm_bw_table.py $O/pmc_hbm_traffic.json $O/kernel_trace_serial.txt > $O/hbm_bandwidth_per_kernel.txt 2>&1
# the configuration's bench line LAST, with the instrumented passes: bench.py looks the counter file up under profiles/ by the configuration's
# tag (config_tag: cfg2 / cfg4), so that roofline.traffic of this line names the file and the build it was just measured on
[ -n "$3" ] && cp $O/pmc_hbm_traffic.json profiles/$3
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $FLAGS 2>$O/bench.err | tail -1 > $O/bench.json
rm -rf $O/serial $O/FETCH_SIZE $O/WRITE_SIZE
cut -c1-300 $O/bench.json; head -22 $O/kernel_trace_serial.txt | cut -c1-170; head -20 $O/hbm_bandwidth_per_kernel.txt
