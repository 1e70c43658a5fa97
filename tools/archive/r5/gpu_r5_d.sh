#!/bin/bash
# r5 call d: whole GPU suite + smoke on the stripped production build (no PROF / atomic / hipGraph paths; one fragment offset per tap shift
# in the shared-tile 3x3 kernel), then the step
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5d; mkdir -p $O
rm -f gpurun_out/parity_report.txt
timeout 1500 python -m pytest tests -m gpu -q -rf --timeout 900 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -v "^$" $O/pytest.log | tail -15
cp gpurun_out/parity_report.txt $O/parity_report.txt 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
for i in 1 2; do
timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-events > $O/bench_$i.json 2> $O/bench_$i.err
python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$i.json").read().strip().splitlines()[-1]); print("run $i: %.3f ms/step median %.3f  %.1f img/s" % (d["ms_per_step"], d["ms_per_step_median_hipevent"], d["value"]))
except Exception as e:
    print("run $i failed", e); print(open("$O/bench_$i.err").read()[-1500:])
PY
done
