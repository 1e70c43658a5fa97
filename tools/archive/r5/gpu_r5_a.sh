#!/bin/bash
# r5 call a: A13 pin — product / oracle masks vs the reference-compiled nms_kernel.cu; baseline bench of the round-4 build on this box
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5a; mkdir -p $O
rm -f gpurun_out/parity_report.txt
timeout 600 python -m pytest tests/test_nms_ref_gpu.py -m gpu -q -rf -p no:cacheprovider > $O/nms_ref.log 2>&1; echo "rc=$?" >> $O/nms_ref.log
tail -5 $O/nms_ref.log; cp gpurun_out/parity_report.txt $O/ 2>/dev/null; cat $O/parity_report.txt
timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-events > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
