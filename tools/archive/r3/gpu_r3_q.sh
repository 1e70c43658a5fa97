#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/gpu_tests.sh
bash tools/gpu_pmc.sh r3pmc_final > gpurun_out/r3pmc_final.log 2>&1; tail -30 gpurun_out/r3pmc_final.log | cut -c1-180
