#!/bin/bash
# the model-level parity tests with every off-by-default path switched ON (kept code must stay correct)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r3z; mkdir -p $O
export MPN_BN_FIN_GROUP_MAX_TILES=16384 MPN_BN_ACT_FINALIZE=1 MPN_FUSE_RELU_BWD=1 MPN_CAT_SPLIT_DGRAD=1 MPN_PW_EPI_MASK=7
timeout 1700 python -m pytest tests/test_model_gpu.py tests/test_replay_gpu.py tests/test_round2_gpu.py -q -m gpu -p no:cacheprovider -k "not finalize" > $O/tests_all_on.log 2>&1; tail -6 $O/tests_all_on.log
