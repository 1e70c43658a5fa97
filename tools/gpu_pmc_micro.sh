#!/bin/bash
# SQ / LDS / TA counters of the layer-3 conv shapes (microbenchmark) — one rocprofv3 --pmc pass per counter group
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s11; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp
i=0
for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU" \
         "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
         "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  MB_ONLY=3,4 MB_WGRAD=1 MB_COLD=1 MB_ITERS=10 timeout 300 rocprofv3 --kernel-trace --pmc $G -d $R/$O/g$i -- python $R/tools/conv_microbench.py > $R/$O/g$i.out 2>&1
  DB=$(find $R/$O/g$i -name "*_results.db" | head -1)
  echo "== group $i: $G" >> $R/$O/pmc_micro.txt
  [ -n "$DB" ] && python $R/tools/pmc_generic.py "$DB" 2>&1 | grep -E "conv_igemm|conv_wgrad|reduce_partials" >> $R/$O/pmc_micro.txt
  [ -z "$DB" ] && tail -5 $R/$O/g$i.out >> $R/$O/pmc_micro.txt
  rm -rf $R/$O/g$i
done
cat $R/$O/pmc_micro.txt | cut -c1-200
