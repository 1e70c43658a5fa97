#!/usr/bin/env python3
"""Timing-only ablation of the recorded step: drop the named C-ABI launches from the recorded launch list and run bench.py.

    MPN_ABLATE_LAUNCHES=mpn_bn_finalize_train,mpn_bn_bwd_finalize python tools/ablate_launches.py --steps 30 --warmup 5 ...

The launches still run while the step is RECORDED (their outputs keep the values of that step — bench.py feeds the same batch every
step, so the numbers downstream stay sane), only the replays skip them.  The figure answers "what would the step take if this work
cost nothing" — the ceiling of any scheme that folds those launches into their neighbours."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multiposenet.pytorch_amd import _lib      # noqa: E402  (before anything binds _lib.call)

SKIP = set(filter(None, os.environ.get("MPN_ABLATE_LAUNCHES", "").split(",")))
_orig = _lib.call
dropped = {"n": 0}


def call(name, *args):
    r = _orig(name, *args)
    if name in SKIP and _lib.TAPE is not None and _lib.TAPE:
        _lib.TAPE.pop()
        dropped["n"] += 1
    return r


_lib.call = call
import torch                                    # noqa: E402
import bench                                    # noqa: E402

if SKIP:       # skipped producers leave pool memory with a previous tenant's bytes: the loss of an ablated run means nothing (and may be NaN)
    torch.isfinite = lambda t: torch.ones_like(t, dtype=torch.bool)

if __name__ == "__main__":
    bench.main()
    sys.stderr.write("ablate_launches: %d recorded launches dropped (%s)\n" % (dropped["n"], ",".join(sorted(SKIP))))
