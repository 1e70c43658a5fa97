#!/bin/bash
# r5 final call: whole GPU suite + smoke on the final build, then the artefacts (PMC first), cfg2 / cfg4 / cfg5 lines
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5final; mkdir -p $O
rm -f gpurun_out/parity_report.txt
timeout 1500 python -m pytest tests -m gpu -q -rf --timeout 900 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -v "^$" $O/pytest.log | tail -4
cp gpurun_out/parity_report.txt $O/parity_report.txt 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
bash tools/gpu_pmc.sh r5final > $O/pmc.log 2>&1
cp $O/pmc_hbm_traffic.json profiles/r05_pmc_hbm_traffic.json
bash tools/gpu_final.sh r5final 2>&1 | tail -12 | cut -c1-200
python tools/hbm_bw_table.py $O/pmc_hbm_traffic.json $O/kernel_trace_serial.txt > $O/hbm_bandwidth_per_kernel.txt 2>&1
timeout 400 python bench.py --layers 50 --size 480 --batch 16 --dtype f32 --subnet keypoint_subnet --steps 20 --warmup 5 --no-cpu-baseline > $O/cfg2_bench.json 2> $O/cfg2.err
python -c "import json; d=json.loads(open('$O/cfg2_bench.json').read().strip().splitlines()[-1]); print('cfg2', d['value'], d['ms_per_step'], d.get('roofline',{}).get('kernel'), d.get('roofline',{}).get('frac'))"
timeout 400 python bench.py --size 800 --batch 8 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events > $O/cfg4_bench.json 2> $O/cfg4.err
python -c "import json; d=json.loads(open('$O/cfg4_bench.json').read().strip().splitlines()[-1]); print('cfg4', d['value'], d['ms_per_step'])"
timeout 600 python tools/infer_bench.py --iters 8 2>/dev/null > $O/cfg5_infer_bench.txt; cut -c1-200 $O/cfg5_infer_bench.txt
