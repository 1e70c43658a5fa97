#!/bin/bash
# round 4, call K: round-4 GPU tests (spawn path, recorded-step cache, TTA, atomic statistics) + the suites touched by the moves
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4k; mkdir -p $O
timeout 1500 python -m pytest tests/test_round4_gpu.py -q -x -m gpu -p no:cacheprovider > $O/tests_round4.log 2>&1; tail -15 $O/tests_round4.log
timeout 1500 python -m pytest tests/test_replay_gpu.py tests/test_harness_gpu.py tests/test_round3_gpu.py -q -x -m gpu -p no:cacheprovider -k "replay or trainer or tester or val or cfg5 or harness" > $O/tests_other.log 2>&1; tail -5 $O/tests_other.log
