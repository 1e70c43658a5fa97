#!/bin/bash
# round 4: the whole GPU suite + smoke() on the final build
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4_suite; mkdir -p $O
timeout 3000 python -m pytest tests/ -q -m gpu -p no:cacheprovider > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log | cut -c1-300
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -3 $O/smoke.log | cut -c1-300
cp gpurun_out/parity_report.txt $O/ 2>/dev/null
