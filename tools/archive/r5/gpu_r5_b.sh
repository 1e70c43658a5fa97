#!/bin/bash
# r5 call b: fused Conv->BN->ReLU launch (MpnConvParams.fz): parity tests, then A/B of the step (MPN_FUSE_BN_ACT 0 / 1, alternating)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5b; mkdir -p $O
rm -f gpurun_out/parity_report.txt
timeout 900 python -m pytest tests/test_round5_gpu.py -m gpu -q -rf -x -p no:cacheprovider --timeout 600 > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -15 $O/tests.log; cp gpurun_out/parity_report.txt $O/ 2>/dev/null; cat $O/parity_report.txt
for i in 1 2; do
  for f in 0 1; do
    MPN_FUSE_BN_ACT=$f timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-events > $O/bench_f${f}_$i.json 2> $O/bench_f${f}_$i.err
    python - <<PY
import json
try:
    d=json.loads(open("$O/bench_f${f}_$i.json").read().strip().splitlines()[-1]); print("fuse=$f run $i: %.3f ms/step median %.3f  %.1f img/s" % (d["ms_per_step"], d["ms_per_step_median_hipevent"], d["value"]))
except Exception as e:
    print("fuse=$f run $i failed", e); print(open("$O/bench_f${f}_$i.err").read()[-1500:])
PY
  done
done
