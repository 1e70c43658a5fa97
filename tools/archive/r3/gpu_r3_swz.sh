#!/bin/bash
# pixel-tile swizzle key of conv_igemm_s3_kernel that is conflict-free at row offsets 0 / 1 / 2: parity, LDS conflict counters on the
# 3x3 micro-benchmark shapes (old vs new library), step A/B (old vs new library, interleaved)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3swz; mkdir -p $O
L=multiposenet/pytorch_amd/libmpn_hip.so
R=$GRAFT_REPO_ROOT
cp tools/libmpn_new.so $L
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_round2_gpu.py tests/test_round3_gpu.py -x -q -m gpu -p no:cacheprovider -k "conv or golden or s3 or pyramid or concat or cfg5 or bottleneck" > $O/tests.log 2>&1; tail -3 $O/tests.log
for V in old new; do
  cp tools/libmpn_$V.so $L
  cd /tmp
  MB_ONLY=4,5,6 MB_WGRAD=0 MB_COLD=1 MB_ITERS=10 timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $R/$O/g_$V -- python $R/tools/conv_microbench.py > $R/$O/g_$V.out 2>&1
  cd $R
  DB=$(find $O/g_$V -name "*_results.db" | head -1)
  echo "== library $V" >> $O/pmc_swz.txt
  [ -n "$DB" ] && python tools/pmc_generic.py "$DB" 2>&1 | grep -E "conv_igemm" >> $O/pmc_swz.txt
  grep -E "3x3" $O/g_$V.out >> $O/pmc_swz.txt
  rm -rf $O/g_$V
done
cat $O/pmc_swz.txt | cut -c1-200
for V in old new old new old new; do
  cp tools/libmpn_$V.so $L
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V', d['ms_per_step_median_hipevent'], d['ms_per_step'], d['value'])"
done | tee $O/ab.txt
cp tools/libmpn_new.so $L
