#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s8; mkdir -p $O


timeout 600 python tools/infer_bench.py > $O/infer_fold.txt 2>&1; cat $O/infer_fold.txt | tail -3
timeout 600 python tools/infer_bench.py --no-fold > $O/infer_nofold.txt 2>&1; tail -2 $O/infer_nofold.txt
timeout 600 python tools/infer_bench.py --dtype bf16 > $O/infer_bf16.txt 2>&1; tail -2 $O/infer_bf16.txt
