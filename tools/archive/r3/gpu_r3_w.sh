#!/bin/bash
# round-end artefacts on the final library build: full GPU suite + smoke, PMC HBM traffic, bench lines, kernel traces, shape reports
cd $GRAFT_REPO_ROOT
bash tools/gpu_tests.sh
bash tools/gpu_pmc.sh r3pmc_final > gpurun_out/r3pmc_final.log 2>&1; tail -12 gpurun_out/r3pmc_final.log | cut -c1-160
bash tools/gpu_final.sh r3final | cut -c1-400
