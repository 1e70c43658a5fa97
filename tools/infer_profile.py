#!/usr/bin/env python3
"""cProfile of the post-processing stage of tools/infer_bench.py (heat-map peaks + PRN assignment), host side."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0], "--iters", "3"]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import infer_bench as ib      # noqa

pr = cProfile.Profile()
pr.enable()
ib.main()
pr.disable()
st = pstats.Stats(pr).sort_stats("cumulative")
st.print_stats("pytorch_amd|infer_bench|tolist|numpy|item|cpu|built-in", 60)
