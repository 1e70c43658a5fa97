#!/bin/bash
mkdir -p gpurun_out/r6c12
for m in 0 1; do
  MPN_CONV2_CLASSES=$m python -m pytest tests/test_round6_gpu.py -x -q -k "training_curves" > gpurun_out/r6c12/test_train_cls$m.log 2>&1
  echo "classes=$m: $(grep 'does it train' gpurun_out/parity_report.txt | tail -1)"
done
python tools/train_sanity.py --steps 200 > gpurun_out/r6c12/train_sanity.txt 2>&1; tail -22 gpurun_out/r6c12/train_sanity.txt
