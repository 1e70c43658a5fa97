"""Round-3 CPU-tier tests: bench.py's rank launching contract, the rewritten Trainer / checkpoint helpers against fixtures
recorded from the REAL reference classes (tests/golden/make_golden_trainer.py), host-side logic added this round."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from helpers import ROOT


def _run(cmd, env=None, timeout=120):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=e, cwd=ROOT)


def test_bench_gpus_flag_is_binding():
    """`--gpus N` is the job size: with fewer visible devices (none here) bench.py fails loudly instead of printing a line
    labelled n_gpus: 1; a launcher whose WORLD_SIZE disagrees with --gpus is refused as well."""
    r = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"], env={"HIP_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": ""})
    assert r.returncode == 3 and "--gpus 2 requested" in r.stderr and "n_gpus" not in r.stdout
    r = _run([sys.executable, "bench.py", "--gpus", "1", "--steps", "1", "--warmup", "0"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 3 and "WORLD_SIZE=2" in r.stderr and "n_gpus" not in r.stdout
    r = _run([sys.executable, "bench.py", "--gpus", "0"])
    assert r.returncode != 0
