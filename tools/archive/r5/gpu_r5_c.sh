#!/bin/bash
# r5 call c: where the fused Conv->BN->ReLU launch loses: layer3 chain as a replayed launch list, ablations of the fused tail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5c; mkdir -p $O
timeout 300 python -m pytest tests/test_round5_gpu.py -m gpu -q -x -p no:cacheprovider --timeout 250 -k "recorded or three_launch" > $O/tests.log 2>&1; tail -3 $O/tests.log
for f in 0 512 1024 2048 4096 1536; do
  MPN_DEBUG_FLAGS=$f timeout 200 python tools/fuse_bn_microbench.py 23 20 2>&1 | tee -a $O/micro.txt
done
MB_HW=15 MB_C=2048 timeout 200 python tools/fuse_bn_microbench.py 3 50 2>&1 | tee -a $O/micro.txt
