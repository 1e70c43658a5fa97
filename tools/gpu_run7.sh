#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s8; mkdir -p $O
timeout 900 python -m pytest tests/test_round2_gpu.py tests/test_model_gpu.py -m gpu -q -rf --timeout 600 -p no:cacheprovider -k "folded or golden or cfg5 or all_images or independence" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -v "^$" $O/pytest.log | tail -15
timeout 600 python tools/infer_bench.py > $O/infer_fold.txt 2>&1; cat $O/infer_fold.txt | tail -3
timeout 600 python tools/infer_bench.py --no-fold > $O/infer_nofold.txt 2>&1; tail -2 $O/infer_nofold.txt
timeout 600 python tools/infer_bench.py --dtype bf16 > $O/infer_bf16.txt 2>&1; tail -2 $O/infer_bf16.txt
