#!/usr/bin/env python3
"""bench.py on libmpn_hip_experiments.so (csrc/Makefile `experiments`): the only way the environment overrides of tuned constants
(MPN_TC*_MIN_*, MPN_WGRAD_TARGET, ...) and the ablation bits reach the kernels.  usage: python tools/bench_experiments.py <bench.py flags>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multiposenet.pytorch_amd import _lib
_lib.use_experiments_build()
import bench
bench.main()
