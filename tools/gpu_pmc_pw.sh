#!/bin/bash
# SQ / TA / TCC counters of conv_pw_kernel vs the generic kernel on 256->1024 @30 (microbenchmark), one rocprofv3 --pmc pass per group
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3pmc; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp
i=0
for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVES" \
         "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
         "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
         "GRBM_GUI_ACTIVE TCP_TCC_WRITE_REQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  MPN_PW_CFG=${PWCFG:-2} PW_ONLY=0 PW_VARIANTS=${PWVAR:-0} timeout 300 rocprofv3 --kernel-trace --pmc $G -d $R/$O/g$i -- python $R/tools/pw_microbench.py > $R/$O/g$i.out 2>&1
  DB=$(find $R/$O/g$i -name "*_results.db" | head -1)
  echo "== group $i: $G" >> $R/$O/pmc_pw.txt
  [ -n "$DB" ] && python $R/tools/pmc_generic.py "$DB" 2>&1 | grep -E "conv_igemm|conv_pw" >> $R/$O/pmc_pw.txt
  [ -z "$DB" ] && tail -5 $R/$O/g$i.out >> $R/$O/pmc_pw.txt
  rm -rf $R/$O/g$i
done
cat $R/$O/pmc_pw.txt | cut -c1-220
