#!/bin/bash
# round 4, call AF: shader clock and power while the step runs (is the chip power-capped under this load?) + the k-loop profile with its span column
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4af; mkdir -p $O
rocm-smi --showclocks --showpower --showmaxpower 2>&1 | grep -v "^$" | head -30 > $O/smi_idle.txt
( for i in $(seq 1 60); do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|fclk\|Power" | tr '\n' ' '; echo; sleep 0.5; done ) > $O/smi_during_bench.txt &
SMI=$!
timeout 300 python bench.py --steps 300 --warmup 8 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | cut -c1-200
kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
head -3 $O/smi_idle.txt; sed -n 20,30p $O/smi_during_bench.txt | cut -c1-400
for v in 0 1; do
  echo "== MPN_WGRAD_S3=$v"
  MPN_WGRAD_S3=$v timeout 600 python tools/kloop_profile.py 2>&1 | grep -A1 "^3x3" | grep "wgrad\|^3x"
done | tee $O/kloop_s3_span.txt
