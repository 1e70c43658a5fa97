"""Round 5 GPU tests: the data-parallel line describes its process group and RCCL's kernels run beside backward."""
import glob
import json
import os
import sqlite3
import subprocess
import sys

import pytest
import torch

from helpers import ROOT, report

pytestmark = pytest.mark.gpu


def test_force_dist_line_names_rccl_and_its_kernels_stay_off_the_main_stream(tmp_path):
    """`bench.py --force-dist` (one rank, RCCL: the most a 1-GPU box can show) under `rocprofv3 --kernel-trace`: the JSON line carries the
    `dist` block (backend nccl, world size as torch.distributed reports it, RCCL version, buckets, per-rank step time, exposed all-reduce
    time), and in the trace every ncclDevKernel launch sits on a queue that is neither the step's main stream — the one with the most
    launches — nor empty of company: the first collective of a step starts while the main stream's backward is still running.
    Replaces the reference's per-step scatter / replicate / gather (datasets/data_parallel.py:16-87)."""
    out_dir = str(tmp_path / "trace")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = ["rocprofv3", "--kernel-trace", "-d", out_dir, "--", sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--steps", "4",
           "--warmup", "2", "--layers", "50", "--size", "256", "--batch", "8", "--no-kernel-events", "--no-cpu-baseline"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd="/tmp")
    assert res.returncode == 0, "rocprofv3 bench.py --force-dist failed (rc %d):\n%s" % (res.returncode, res.stderr[-3000:])
    line = [ln for ln in res.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(line) == 1, res.stdout[-2000:]
    out = json.loads(line[0])
    d = out["dist"]
    assert d["backend"] == "nccl" and d["world_size"] == 1 and d["rccl_version"] not in (None, "unavailable"), d
    assert d["buckets"] >= 2 and d["collectives_per_step"] == d["buckets"] and len(d["per_rank_ms"]) == 1
    assert d["allreduce_ms_exposed"][0] is not None and 0.0 <= d["allreduce_ms_exposed"][0] < out["ms_per_step"]
    dbs = glob.glob(os.path.join(out_dir, "**", "*_results.db"), recursive=True)
    assert dbs, "no rocpd database under %s" % out_dir
    cur = sqlite3.connect(dbs[0]).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    qcol = next((c for c in ("stream_id", "queue_id", "queue", "stream") if c in cols), None)
    assert qcol is not None, cols
    rows = cur.execute("select start, end, name, %s from kernels order by start" % qcol).fetchall()
    counts = {}
    for r in rows:
        counts[r[3]] = counts.get(r[3], 0) + 1
    main_q = max(counts, key=counts.get)
    nccl = [r for r in rows if "nccl" in r[2].lower()]
    assert len(nccl) >= d["buckets"] * 4, "expected >= %d collective kernels in the trace, found %d" % (d["buckets"] * 4, len(nccl))
    on_main = [r for r in nccl if r[3] == main_q]
    assert not on_main, "%d RCCL kernels were enqueued on the step's main stream" % len(on_main)
    # overlap: per optimizer step, the first collective starts before the main stream's last pre-optimizer kernel has ended
    adam = [r for r in rows if "adam" in r[2]]
    overlapped = 0
    for a, b in zip(adam[:-1], adam[1:]):
        step = [r for r in rows if a[1] <= r[0] and r[1] <= b[0]]
        m = [r for r in step if r[3] == main_q]
        c = [r for r in step if "nccl" in r[2].lower()]
        if m and c and min(r[0] for r in c) < max(r[1] for r in m):
            overlapped += 1
    assert overlapped >= 2, "no step shows a collective in flight while the main stream still runs backward"
    report("bench.py --force-dist under rocprofv3: dist block %s; %d RCCL kernels, none on the main queue (%d launches there), %d steps with a collective "
           "in flight beside backward; exposed all-reduce %.3f ms of %.2f ms/step"
           % ({k: d[k] for k in ("backend", "world_size", "rccl_version", "buckets")}, len(nccl), counts[main_q], overlapped,
              d["allreduce_ms_exposed"][0], out["ms_per_step"]))
