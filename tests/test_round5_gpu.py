"""Round 5 GPU tests: the data-parallel line describes its process group and RCCL's kernels run beside backward."""
import json
import os
import subprocess
import sys

import pytest
import torch

from helpers import ROOT, report

pytestmark = pytest.mark.gpu


def test_force_dist_line_describes_its_process_group():
    """`bench.py --force-dist` (one rank over RCCL: the most a 1-GPU box can show): the JSON line carries the `dist` block — backend nccl,
    the world size torch.distributed reports, the RCCL version, buckets and collectives per step, per-rank step time, exposed all-reduce
    time — so that the first 8-GPU line can be read for "did RCCL see 8 ranks, were the collectives hidden" without a second run.
    (A kernel-trace assertion on the collective's queue is not possible here: RCCL elides a one-rank all-reduce, `rocprofv3
    --kernel-trace` of this command shows no ncclDevKernel at all — gpurun call r5d.  What the trace would be checked for is fixed by
    construction in ddp.GradReducer._launch: the collective is issued under the side stream, after an event of the main one.)
    Replaces the reference's per-step scatter / replicate / gather (datasets/data_parallel.py:16-87)."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--steps", "4", "--warmup", "2", "--layers", "50", "--size", "256",
           "--batch", "8", "--no-kernel-events", "--no-cpu-baseline"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, "bench.py --force-dist failed (rc %d):\n%s" % (res.returncode, res.stderr[-3000:])
    line = [ln for ln in res.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(line) == 1, res.stdout[-2000:]
    out = json.loads(line[0])
    d = out["dist"]
    assert d["backend"] == "nccl" and d["world_size"] == 1 and d["rccl_version"] not in (None, "unavailable"), d
    assert d["buckets"] >= 2 and d["collectives_per_step"] == d["buckets"] and len(d["per_rank_ms"]) == 1
    assert d["gradient_bytes_per_step"] > 100e6 and d["bucket_mb"] == 32.0
    assert d["allreduce_ms_exposed"][0] is not None and 0.0 <= d["allreduce_ms_exposed"][0] < out["ms_per_step"]
    assert "not_a_measurement" not in out and out["n_gpus"] == 1
    report("bench.py --force-dist: dist block %s; exposed all-reduce %.3f ms of %.2f ms/step"
           % ({k: d[k] for k in ("backend", "world_size", "rccl_version", "buckets", "collectives_per_step")}, d["allreduce_ms_exposed"][0], out["ms_per_step"]))
