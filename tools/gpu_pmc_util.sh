#!/bin/bash
# MFMA / LDS utilisation per kernel of the training step: rocprofv3 --pmc over the bench command (serial schedule, 3 timed steps),
# one pass per counter group; summary -> gpurun_out/$1/pmc_utilisation.txt
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
TAG=${1:-pmcutil}; O=gpurun_out/$TAG; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp
i=0
for G in "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" \
         "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  MPN_SIDE_STREAM=0 timeout 900 rocprofv3 --kernel-trace --pmc $G -d $R/$O/g$i -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events > $R/$O/g$i.out 2>&1
  DB=$(find $R/$O/g$i -name "*_results.db" | head -1)
  echo "== group $i: $G" >> $R/$O/pmc_raw.txt
  [ -n "$DB" ] && python $R/tools/pmc_generic.py "$DB" >> $R/$O/pmc_raw.txt 2>&1
  [ -z "$DB" ] && tail -5 $R/$O/g$i.out >> $R/$O/pmc_raw.txt
  rm -rf $R/$O/g$i
done
cd $R
python tools/pmc_util_table.py $O/pmc_raw.txt > $O/pmc_utilisation.txt 2>&1; head -40 $O/pmc_utilisation.txt | cut -c1-200
