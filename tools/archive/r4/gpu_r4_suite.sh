#!/bin/bash
# round 4: the whole GPU suite + smoke() on the final build; then the model / replay / round-2 suites with the off-by-default atomic BatchNorm statistics ON
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4_suite; mkdir -p $O; rm -f gpurun_out/parity_report.txt
timeout 3000 python -m pytest tests/ -q -m gpu -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log | cut -c1-300
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -3 $O/smoke.log | cut -c1-300
cp gpurun_out/parity_report.txt $O/ 2>/dev/null
MPN_BN_ATOMIC_STATS=1 timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_round2_gpu.py tests/test_round3_gpu.py -q -m gpu -p no:cacheprovider > $O/pytest_gpu_atomic_on.log 2>&1; tail -2 $O/pytest_gpu_atomic_on.log | cut -c1-300
MPN_DGRAD_S2_CLASSES=0 MPN_WGRAD_LIN=0 timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_round2_gpu.py -q -m gpu -p no:cacheprovider > $O/pytest_gpu_r4_paths_off.log 2>&1; tail -2 $O/pytest_gpu_r4_paths_off.log | cut -c1-300
