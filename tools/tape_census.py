#!/usr/bin/env python3
"""What is on the recorded launch list of one training step?  Entry points by count (python tools/tape_census.py, GPU box)."""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from multiposenet.pytorch_amd import replay


def main():
    sys.argv = ["bench.py", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-kernel-events"]
    seen = {}
    orig = replay.ReplayedTrainStep._record

    def rec(self, *a, **k):
        ent = orig(self, *a, **k)
        seen["tape"] = ent.tape
        return ent
    replay.ReplayedTrainStep._record = rec
    try:
        bench.main()
    except SystemExit:
        pass
    tape = seen.get("tape") or []
    c = collections.Counter()
    for item in tape:
        name = item[0] if isinstance(item, (tuple, list)) else getattr(item, "name", str(type(item)))
        if not isinstance(name, str):
            name = getattr(name, "__name__", None) or getattr(name, "name", None) or str(name)
        c[name] += 1
    print("entries on the tape:", len(tape))
    for n, k in c.most_common(60):
        print("%5d  %s" % (k, n))


if __name__ == "__main__":
    main()
