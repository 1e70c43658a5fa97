#!/bin/bash
# round 4 artefacts on the final library build: bench line (+ cpu_baseline), 1-rank RCCL line, serial / overlapped kernel traces, serial
# shape report, PMC HBM traffic, PMC utilisation
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
bash tools/gpu_final.sh r4_final
bash tools/gpu_pmc.sh r4_final_pmc
bash tools/gpu_pmc_util.sh r4_final_util
python tools/hbm_bw_table.py gpurun_out/r4_final_pmc/pmc_hbm_traffic.json gpurun_out/r4_final/kernel_trace_serial.txt > gpurun_out/r4_final_pmc/hbm_bandwidth_per_kernel.txt
head -30 gpurun_out/r4_final_pmc/hbm_bandwidth_per_kernel.txt
