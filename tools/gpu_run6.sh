#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s6; mkdir -p $O
timeout 900 python -m pytest tests/test_round2_gpu.py tests/test_model_gpu.py tests/test_replay_gpu.py -m gpu -q -rf --timeout 600 -p no:cacheprovider -k "pyramid or golden or replay or deterministic or bn_backward_statistics or graphed" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -v "^$" $O/pytest.log | tail -30
for P in 1 0; do
  MPN_BN_FUSED_STATS=$P timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events > $O/bench_p$P.json 2> $O/bench_p$P.err; echo "bnfused=$P rc=$? $(python -c "import json;d=json.load(open('$O/bench_p$P.json'));print(d['value'],d['ms_per_step'],d['ms_per_step_median_hipevent'])")"; tail -2 $O/bench_p$P.err
done
du -sh $O
