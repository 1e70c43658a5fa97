#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s3; mkdir -p $O
timeout 1500 python -m pytest tests/test_harness_gpu.py tests/test_prn_assign.py tests/test_peaks_gpu.py -m gpu -q -rf --timeout 900 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -v "^$" $O/pytest.log | tail -70
cp gpurun_out/parity_report.txt $O/ 2>/dev/null
du -sh $O
