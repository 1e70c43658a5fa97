#!/bin/bash
# r5 final artefacts on the final build: PMC HBM traffic FIRST (so that the bench line's traffic_source names this build), then the default
# bench line (with cpu_baseline), the 1-rank RCCL line, serial / overlap kernel traces (+ per-step census of non-library launches), shape report
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5final; mkdir -p $O
bash tools/gpu_pmc.sh r5final > $O/pmc.log 2>&1
cp $O/pmc_hbm_traffic.json profiles/r05_pmc_hbm_traffic.json        # bench.py reads the newest profiles/rNN_pmc_hbm_traffic.json
bash tools/gpu_final.sh r5final 2>&1 | tail -40
python tools/hbm_bw_table.py $O/pmc_hbm_traffic.json $O/kernel_trace_serial.txt > $O/hbm_bandwidth_per_kernel.txt 2>&1; head -12 $O/hbm_bandwidth_per_kernel.txt | cut -c1-180
