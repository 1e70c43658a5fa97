#!/bin/bash
# final build, counter files of this build already under profiles/: the bench lines that cite them
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6c20; mkdir -p $O
for rep in 1 2 3; do for m in 0 1; do
  MPN_CONV2_CLASSES=$m python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-events > $O/bench_cls${m}_$rep.json 2> $O/bench_cls${m}_$rep.err
  echo "conv2 classes=$m rep $rep: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' $O/bench_cls${m}_$rep.json | tr '\n' ' ')"
done; done
python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-160 $O/bench_default.json
python bench.py --layers 50 --size 480 --batch 16 --dtype f32 --subnet keypoint_subnet --steps 20 --warmup 5 --no-cpu-baseline > $O/cfg2_bench.json 2> $O/cfg2.err; cut -c1-160 $O/cfg2_bench.json
python bench.py --layers 101 --size 800 --batch 8 --steps 20 --warmup 5 --no-cpu-baseline > $O/cfg4_bench.json 2> $O/cfg4.err; cut -c1-160 $O/cfg4_bench.json
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof -- python $R/tools/infer_bench.py --iters 4 > $R/$O/prof.out 2>&1
cd $R
DB=$(find $O/prof -name "*_results.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py "$DB" 44 "round 6 (final build), cfg5: python tools/infer_bench.py --iters 4 (R101 both 640x640 B=64 f16, BN folded; 5 serial stage sets x (1 + 4) batches + 1 + 3 + 12 + 3 pipelined batches = 44 batches; calibration passes included), rocprofv3 --kernel-trace --stats" > $O/cfg5_kernel_trace.txt 2>&1
rm -rf $O/prof
head -16 $O/cfg5_kernel_trace.txt | cut -c1-170
