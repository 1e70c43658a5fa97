#!/bin/bash
# round-6 end artefacts, part 1: whole GPU suite + smoke, PMC traffic of the headline FIRST, then bench lines / traces / shape report
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp MPN_ROUND=6
O=gpurun_out/r6final; mkdir -p $O
rm -f gpurun_out/parity_report.txt
timeout 1700 python -m pytest tests -m gpu -q -rf --timeout 900 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -v "^$" $O/pytest.log | tail -4
cp gpurun_out/parity_report.txt $O/parity_report.txt 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
bash tools/gpu_pmc.sh r6final > $O/pmc.log 2>&1
cp $O/pmc_hbm_traffic.json profiles/r06_pmc_hbm_traffic.json        # bench.py reads the newest profiles/rNN_pmc_hbm_traffic.json
bash tools/gpu_final.sh r6final 2>&1 | tail -14 | cut -c1-220
python tools/hbm_bw_table.py $O/pmc_hbm_traffic.json $O/kernel_trace_serial.txt > $O/hbm_bandwidth_per_kernel.txt 2>&1; head -12 $O/hbm_bandwidth_per_kernel.txt | cut -c1-180
