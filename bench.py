#!/usr/bin/env python3
"""bench.py — images/sec of the MultiPoseNet training step (fwd + losses + bwd + Adam) on MI355X.

Workload (BASELINE.json metric / SURVEY.md 8d cfg3): ResNet-101 full posenet (shared backbone, both
pyramids, keypoint head + heat-map MSE, RetinaNet heads + focal/smooth-L1), 480x480, 32 images/GPU,
bf16 MFMA arithmetic with fp32 accumulation and fp32 master weights, synthetic COCO-shaped batches
resident in HBM.  One process per GPU; N>1 = data parallel (RCCL all-reduce of the flat gradient
arena overlapped with backward); weak scaling (32 images per GPU).

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...        # no launcher: re-executes itself under torch.distributed.run with N ranks

`n_gpus` in the result always equals --gpus: a launcher whose WORLD_SIZE disagrees, or fewer visible devices than ranks, is an
error (exit status 3, no JSON line), never a silently smaller job.

The timed step is the recorded step of multiposenet/pytorch_amd/replay.py (forward, losses, zero_grad, backward on two
HIP streams, RCCL buckets, Adam — recorded once as a launch list and re-issued per step (replay.py); `--launch eager`
times the Python tape instead).  Every timed
step is also bracketed by a pair of HIP events on the launch stream; their median is reported beside the wall-clock
mean that `value` is computed from.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel class, timed with HIP events on the launch stream
in separate instrumented eager steps AFTER the timed region, one kernel on the GPU at a time (top level); the same
brackets with the weight-gradient side stream on are nested under `overlapped`.  `cpu_baseline` is the oracle's
torch-CPU restatement of the same step timed on this host (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")     # one hardware queue per stream (main / wgrad side stream / RCCL), see the package __init__
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (RCCL across processes); must be in place BEFORE the HSA runtime initialises

import numpy as np
import torch

# BASELINE.md section 3 (algorithmic, 3 x fwd): (layers, size, subnet) -> GFLOP per image of one training step
GFLOP_PER_IMG_TRAIN = {(101, 480, "train_both"): 608.6, (50, 480, "keypoint_subnet"): 343.7, (101, 800, "train_both"): 1690.2,
                       (50, 256, "keypoint_subnet"): 97.8}
PEAK_BF16_TFLOPS = 2500.0     # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--size", type=int, default=480)
    ap.add_argument("--layers", type=int, default=101)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--subnet", default="train_both", choices=["train_both", "keypoint_subnet", "detection_subnet"],
                    help="train_both = the headline step (BASELINE config 3 / 4); keypoint_subnet = BASELINE config 2's step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--eager-log", action="store_true", help="plain-float loss logs (one host sync per step, the reference's behaviour)")
    ap.add_argument("--launch", default="replay", choices=["replay", "eager"],
                    help="replay: recorded launch list (replay.py, default); eager: the Python tape")
    ap.add_argument("--instr-steps", type=int, default=2, help="instrumented eager steps per schedule after the timed region")
    ap.add_argument("--cpu-batch", type=int, default=2)
    ap.add_argument("--cpu-baseline-worker", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL and the gradient reducer even with one rank")
    ap.add_argument("--rccl-channels", type=int, default=0,
                    help="N > 0: NCCL_MIN_NCHANNELS = NCCL_MAX_NCHANNELS = N before the process group comes up (each channel is a "
                         "workgroup taken from the compute kernels); 0 = the library's choice.  Echoed in the line's dist block")
    ap.add_argument("--init-timeout", type=int, default=120, help="seconds a rank waits for rendezvous / the first collective")
    ap.add_argument("--shared-device-test", action="store_true",
                    help="TEST SWITCH, NOT A MEASUREMENT: run the N ranks of --gpus N on ONE device over gloo (RCCL refuses two ranks on one "
                         "GPU), so that the spawn -> rendezvous -> data-parallel step -> one-JSON-line path executes on a 1-GPU box; the line "
                         "carries \"not_a_measurement\": true")
    return ap.parse_args()


def synth(batch, size, device, seed):
    from multiposenet.pytorch_amd import synthetic as weightgen          # data generator only (deterministic Philox), not arithmetic
    img = torch.from_numpy(weightgen.gen_images(seed, batch, size, size))
    heat, wgt = weightgen.gen_keypoint_gt(seed, batch, size // 4, size // 4)
    anno = torch.from_numpy(weightgen.gen_boxes_gt(seed, batch, size, max_n=8))
    return img.to(device), torch.from_numpy(heat).to(device), torch.from_numpy(wgt).to(device), anno.to(device)


def he_weights(model):
    from multiposenet.pytorch_amd import synthetic as weightgen
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = weightgen.gen_state_dict(shapes, seed=0, flavour="he", skip_prefixes=("prn.",))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    return sd


def host_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline_worker(args):
    """Runs in a subprocess (hard timeout in the parent): the oracle's torch-CPU restatement of the
    same step (fwd + both losses + bwd + Adam).  Bounded: a 160x160 calibration step decides whether
    the full 480x480 sample fits the time box; otherwise the calibration is reported, scaled by pixels."""
    from oracle import posenet_oracle as po
    from multiposenet.pytorch_amd import synthetic as weightgen
    cores = min(host_cores(), 64)
    torch.set_num_threads(cores)
    g = np.load(os.path.join(ROOT, "tests", "golden", "g0_keys.npz"))      # state_dict names/shapes of the reference
    shapes = {str(k): tuple(int(v) for v in str(sh).split(",")) if str(sh) else ()
              for k, sh in zip(g["keys_%d" % args.layers], g["shapes_%d" % args.layers])}
    sd = weightgen.gen_state_dict(shapes, seed=0, flavour="he", skip_prefixes=("prn.",))
    params = {k: torch.from_numpy(v).clone() for k, v in sd.items() if v.dtype != np.int64}
    leaves = []
    for k, v in params.items():
        if not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
            leaves.append(v)
    opt = torch.optim.Adam(leaves, lr=1e-4)

    def run(B, S, warm, nsteps):
        img = torch.from_numpy(weightgen.gen_images(1, B, S, S))
        heat, wgt = weightgen.gen_keypoint_gt(1, B, S // 4, S // 4)
        anno = torch.from_numpy(weightgen.gen_boxes_gt(1, B, S, max_n=8))
        ts = []
        for i in range(warm + nsteps):
            t0 = time.time()
            pred, (ks, ds) = po.posenet_forward(params, img, "train_both", args.layers, True)
            l1, _ = po.keypoint_loss(ks, torch.from_numpy(heat), torch.from_numpy(wgt))
            l2, _ = po.detection_loss(ds, anno)
            opt.zero_grad()
            (l1 + l2).backward()
            opt.step()
            ts.append(time.time() - t0)
        ts = sorted(ts[warm:])          # the first steps pay oneDNN primitive creation
        return ts[len(ts) // 2] if len(ts) % 2 else 0.5 * (ts[len(ts) // 2 - 1] + ts[len(ts) // 2])

    # 2 warm-up + 5 timed steps, median.  A 160x160 calibration step decides whether that protocol at the
    # full size fits the time box (7 steps + slack <= 150 s); otherwise fewer timed steps, never fewer than one, and said so.
    B = args.cpu_batch
    t_small = run(B, 160, 1, 1)
    est_full = t_small * (args.size / 160.0) ** 2
    if est_full * 7.5 <= 150.0:
        warm, nsteps = 2, 5
    elif est_full * 3.3 <= 150.0:
        warm, nsteps = 1, 2
    else:
        warm, nsteps = 0, 0
    if nsteps:
        t_full = run(B, args.size, warm, nsteps)
        out = {"value": round(B / t_full, 4), "unit": "images/sec", "cores": cores, "kind": "port", "steps": nsteps, "warmup": warm,
               "sample": "oracle torch-CPU restatement, R%d train_both %dx%d, batch %d, %d warm-up + %d timed steps (fwd+losses+bwd+Adam), "
                         "median %.3f s/step, fp32, %d threads" % (args.layers, args.size, args.size, B, warm, nsteps, t_full, cores)}
    else:
        out = {"value": round(B / est_full, 4), "unit": "images/sec", "cores": cores, "kind": "port", "steps": 1, "warmup": 1,
               "sample": "oracle torch-CPU restatement, R%d train_both at 160x160 batch %d (%.1f s/step), scaled by pixel count to %dx%d; "
                         "the full-size step would exceed the time box; fp32, %d threads" % (args.layers, B, t_small, args.size, args.size, cores)}
    print("CPU_BASELINE " + json.dumps(out), flush=True)


def cpu_baseline(args):
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "--layers", str(args.layers), "--size", str(args.size),
           "--cpu-batch", str(args.cpu_batch)]
    try:
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
        for line in res.stdout.splitlines():
            if line.startswith("CPU_BASELINE "):
                return json.loads(line[len("CPU_BASELINE "):])
        return {"value": None, "unit": "images/sec", "cores": host_cores(), "kind": "port", "sample": "worker failed: %s" % res.stderr[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "images/sec", "cores": host_cores(), "kind": "port", "sample": "timed out after 240 s"}


def config_tag(args):
    """'' for the headline configuration (BASELINE config 3), 'cfg2' / 'cfg4' for the flags that name those configurations, else None."""
    key = (args.layers, args.size, args.batch, args.dtype, args.subnet)
    return {(101, 480, 32, "bf16", "train_both"): "", (50, 480, 16, "f32", "keypoint_subnet"): "cfg2",
            (101, 800, 8, "bf16", "train_both"): "cfg4"}.get(key)


def pmc_traffic(kernel_class, tag=""):
    """HBM bytes per launch of the dominant kernel from the last committed rocprofv3 --pmc passes
    (profiles/r*_pmc_hbm_traffic.json, r*_cfg2_… / r*_cfg4_… for those configurations; made by tools/pmc_summary.py: FETCH_SIZE and
    WRITE_SIZE in separate runs, reads doubled per MI355X_MICROARCH.md).  PMC collection serialises kernels, so it cannot run
    inside the timed bench; None when no profile of THIS configuration is present."""
    import glob
    if tag is None:
        return None, None
    pat = "r[0-9][0-9]_%spmc_hbm_traffic.json" % (tag + "_" if tag else "")
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pat)))
    if not files:
        return None, None
    try:
        prof = json.load(open(files[-1]))
        # the class name ops.py builds for the igemm kernels carries five template arguments, the trace's seven (..., EXT, PROF): exact
        # match first, then the standard (non-extended) instantiation of the same five
        hits = [k for k in prof["kernels"] if k["kernel"] == kernel_class] or \
               [k for k in prof["kernels"] if k["kernel"] == kernel_class[:-1] + ", false, false>"]
        if hits:
            return int(hits[0]["hbm_bytes_per_launch"]), "%s (library build %s)" % (os.path.basename(files[-1]), prof.get("build_id", "unrecorded"))
        return None, None
    except Exception:
        return None, None


def spawn_ranks(args):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks here (one process per GPU under
    torch.distributed.run, rendezvous on 127.0.0.1) — the same command line the driver uses for N > 1.  Never returns."""
    import socket
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if args.shared_device_test and ndev >= 1:
        ndev = args.gpus                                    # every rank will use device 0 (gloo)
    if ndev < args.gpus:
        sys.stderr.write("bench.py: --gpus %d requested but only %d GPU(s) are visible; refusing to run a smaller job under that label\n"
                         % (args.gpus, ndev))
        sys.exit(3)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this host driver (RCCL needs it)
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def main():
    args = parse()
    if args.cpu_baseline_worker:
        cpu_baseline_worker(args)
        return
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks; n_gpus would mislabel the run\n" % (args.gpus, world))
        sys.exit(3)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries exactly one JSON line: RCCL prints a version banner to fd 1 when its first communicator comes up,
    # so fd 1 points at stderr until the result line is written
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    if args.shared_device_test:
        local = 0
    if torch.cuda.device_count() <= local:
        sys.stderr.write("bench.py: rank %d wants cuda:%d but only %d GPU(s) are visible\n" % (rank, local, torch.cuda.device_count()))
        sys.exit(3)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.rccl_channels > 0:
            os.environ["NCCL_MIN_NCHANNELS"] = os.environ["NCCL_MAX_NCHANNELS"] = str(args.rccl_channels)
        import datetime
        tmo = datetime.timedelta(seconds=max(10, args.init_timeout))
        try:
            if args.shared_device_test:
                dist.init_process_group("gloo", timeout=tmo)
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=tmo)
            probe = torch.ones(1, device="cpu" if args.shared_device_test else torch.device("cuda", local))
            dist.all_reduce(probe)                    # the first collective builds the communicator: fail HERE, with a diagnosis
            if not args.shared_device_test:
                torch.cuda.synchronize()
            if int(probe.item()) != world:
                raise RuntimeError("first all-reduce summed to %d over %d ranks" % (int(probe.item()), world))
        except Exception as e:      # noqa: BLE001 — any rendezvous / communicator failure: one line, exit status 3, no JSON
            sys.stderr.write("bench.py: rank %d/%d could not join the process group within %d s (MASTER_ADDR=%s MASTER_PORT=%s, backend %s, "
                             "HSA_ENABLE_IPC_MODE_LEGACY=%s, NCCL_DEBUG=%s): %s: %s\n"
                             % (rank, world, max(10, args.init_timeout), os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT"),
                                "gloo" if args.shared_device_test else "nccl", os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
                                os.environ.get("NCCL_DEBUG", "unset"), type(e).__name__, str(e).splitlines()[0][:300] if str(e) else ""))
            sys.stderr.flush()
            os._exit(3)
    else:
        dist = None
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local if use_dist else 0)

    from multiposenet.pytorch_amd import ddp, ops
    from multiposenet.pytorch_amd.network.posenet import poseNet
    from multiposenet.pytorch_amd.optim import FusedAdam
    from multiposenet.pytorch_amd.network import losses as mpn_losses
    mpn_losses.set_lazy_log(not args.eager_log)     # the log dict is not read inside the timed loop (trainer.py reads it after step())

    cdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    model = poseNet(args.layers, compute_dtype=cdt).to(dev)
    sd = he_weights(model)
    for p in model.prn.parameters():          # PRN is not part of this step (all non-PRN params trainable)
        p.requires_grad = False
    model.train()
    reducer = None
    if use_dist:
        reducer = ddp.attach(model, bucket_mb=32.0)
        reducer.measure = True
    opt = FusedAdam(model, lr=1e-4, weight_decay=0.0)
    img, heat, wgt, anno = synth(args.batch, args.size, dev, seed=100 + rank)

    last_log = {}
    from multiposenet.pytorch_amd.replay import ReplayedTrainStep
    from multiposenet.pytorch_amd.training.batch_processor import train_step
    inputs = [[img, args.subnet]]
    gts = {"train_both": ["train_both", heat, wgt, anno], "keypoint_subnet": ["keypoint_subnet", heat, wgt],
           "detection_subnet": ["detection_subnet", anno]}[args.subnet]
    gstep = ReplayedTrainStep(model, opt) if args.launch == "replay" else None

    def step():
        loss, log = gstep(inputs, gts) if gstep is not None else train_step(model, opt, inputs, gts)
        last_log["log"] = log
        return loss

    if gstep is not None:
        # set-up, not warm-up: the eager passes that fill host-side caches, then the capture + first replay (the
        # equivalent of compiling); the W warm-up steps below are replays like the timed ones
        for _ in range(gstep.eager_steps + 1):
            step()
    for _ in range(args.warmup):
        step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        loss = step()
        marks[i + 1].record()
    t_enq = time.perf_counter() - t0             # host time to enqueue the K steps (the GPU may still be running)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    dist_info = None
    if dist is not None:
        # what the line needs for a reader to see that the collective library really saw `world` ranks
        per_rank = [None] * dist.get_world_size()
        dist.all_gather_object(per_rank, {"rank": rank, "ms_per_step": round(elapsed / args.steps * 1000.0, 3),
                                          "allreduce_ms_exposed": None if reducer.exposed_ms() is None else round(reducer.exposed_ms(), 3)})
        backend = dist.get_backend()
        try:
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None
        except Exception:           # noqa: BLE001 — a build without the binding: say so instead of failing the bench
            rccl = "unavailable"
        dist_info = {"backend": backend, "world_size": dist.get_world_size(), "rccl_version": rccl,
                     "rccl_channels": args.rccl_channels if args.rccl_channels > 0 else "library default",
                     "env": {k: os.environ.get(k, "unset") for k in ("NCCL_MIN_NCHANNELS", "NCCL_MAX_NCHANNELS", "NCCL_DEBUG", "NCCL_ALGO",
                                                                      "NCCL_PROTO", "HSA_ENABLE_IPC_MODE_LEGACY", "GPU_MAX_HW_QUEUES", "MPN_BUCKET_MB",
                                                                      "MPN_BUCKET_ADAM")},
                     "buckets": len(reducer.buckets), "bucket_mb": reducer.bucket_mb, "collectives_per_step": reducer.launched,
                     "optimizer_updates_behind_buckets": reducer.updated,
                     "gradient_bytes_per_step": int(sum(b["end"] - b["start"] for b in reducer.buckets) * 4),
                     "per_rank_ms": [r["ms_per_step"] for r in sorted(per_rank, key=lambda r: r["rank"])],
                     "allreduce_ms_exposed": [r["allreduce_ms_exposed"] for r in sorted(per_rank, key=lambda r: r["rank"])],
                     "allreduce_ms_exposed_how": "HIP events on the launch stream around the waits of GradReducer.finish() in the last timed "
                                                 "step: GPU time between the end of backward and the completion of the last collective AND "
                                                 "of the last bucket's Adam update (per-bucket updates run behind each all-reduce on the "
                                                 "finishing stream)"}
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.shared_device_test else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(loss).all()
    # Host cost of issuing one step, measured with an EMPTY queue (synchronise, enqueue one step, read the clock before the GPU
    # finishes): the in-region figure above is throttled to the GPU's pace once the queue is full.
    host_free = []
    for _ in range(5):
        torch.cuda.synchronize()
        h0 = time.perf_counter()
        step()
        host_free.append(time.perf_counter() - h0)
    torch.cuda.synchronize()
    host_free.sort()
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    median_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])

    # Kernel-level roofline: instrumented EAGER steps outside the timed region (per-launch event brackets serialise
    # neighbouring kernels, and events cannot be read out of a graph replay).  First with the weight-gradient side
    # stream off (one kernel on the GPU at a time: the kernel's own quality, comparable with a serial rocprofv3 trace),
    # then with the production schedule (brackets time-share the GPU with the other stream).
    ke, ke_serial, ev_steps, serial_steps = {}, {}, 0, 0
    if not args.no_kernel_events:
        def eager():
            return train_step(model, opt, inputs, gts)[0]
        overlap = model._engine.overlap_wgrad
        model._engine.overlap_wgrad = False
        eager()                                   # untimed: workspaces of the eager streams
        ops.KERNEL_EVENTS.enable()
        serial_steps = max(1, args.instr_steps)
        for _ in range(serial_steps):
            eager()
        torch.cuda.synchronize()
        ops.KERNEL_EVENTS.disable()
        ke_serial = ops.KERNEL_EVENTS.summary()
        model._engine.overlap_wgrad = overlap
        if overlap:
            eager()
            ops.KERNEL_EVENTS.enable()
            ev_steps = max(1, args.instr_steps)
            for _ in range(ev_steps):
                eager()
            torch.cuda.synchronize()
            ops.KERNEL_EVENTS.disable()
            ke = ops.KERNEL_EVENTS.summary()

    if rank == 0:
        ms = elapsed / args.steps * 1000.0
        ips = args.batch * world * args.steps / elapsed
        out = {
            "metric": "images/sec (train fwd+bwd) ResNet%d %dx%d bs=%d/GPU" % (args.layers, args.size, args.size, args.batch),
            "value": round(ips, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "R%d %s train step: fwd + %s + bwd + Adam, "
                                   "%dx%d, %d images/GPU, %s MFMA / fp32 accumulate / fp32 master weights"
                                   % (args.layers, {"train_both": "full posenet (keypoint+detection)", "keypoint_subnet": "keypoint subnet",
                                                    "detection_subnet": "detection subnet"}[args.subnet],
                                      {"train_both": "MSE/focal losses", "keypoint_subnet": "MSE heat-map loss", "detection_subnet": "focal loss"}[args.subnet],
                                      args.size, args.size, args.batch, args.dtype),
                       "global_batch": args.batch * world, "parallelism": "dp%d" % world,
                       "loss_log": "eager floats (host sync per step)" if args.eager_log else "asynchronous (set_lazy_log)",
                       "launch": {"eager": "eager Python tape (autograd node + ~2300 ctypes launches built per step)",
                                  "replay": "recorded launch list re-issued per step (replay.py)"}[args.launch]
                                 + ("" if gstep is None else ", %d replays" % gstep.replays)},
        }
        out["config"]["optimizer_update"] = ("Adam bucket by bucket behind each all-reduce (finishing stream)" if (gstep is not None and gstep.bucketed_update)
                                             else "Adam in one launch after backward")
        out["config"]["conv2"] = ("keypoint head's conv2 by position classes (x8 / x4 members at their own resolution)" if model._engine.conv2_classes
                                  else "conv2 over the virtual 512-channel concatenation")
        if dist_info is not None:
            out["dist"] = dist_info
        if args.shared_device_test:
            out["not_a_measurement"] = True
            out["config"]["parallelism"] += " (TEST: %d ranks sharing one device over gloo — exercises the launch path, measures nothing)" % world
        out["ms_per_step_median_hipevent"] = round(median_ms, 3)
        out["ms_per_step_min_max_hipevent"] = [round(step_ms[0], 3), round(step_ms[-1], 3)]
        out["last_step_log"] = {k: round(float(v), 6) for k, v in last_log["log"].items()      # values exist, they were just
                                if k in ("heatmap_loss", "total_loss", "classification_loss", "regression_loss")}   # not waited for
        out["host_enqueue_ms_per_step"] = round(host_free[len(host_free) // 2] * 1000.0, 3)      # empty queue, median of 5
        out["host_enqueue_ms_per_step_in_region"] = round(t_enq / args.steps * 1000.0, 3)       # queue full: follows the GPU
        gf = GFLOP_PER_IMG_TRAIN.get((args.layers, args.size, args.subnet))
        if gf is not None:
            out["model_tflops_per_gpu"] = round(ips / world * gf / 1000.0, 2)
        if ke_serial:
            peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS
            name, d = max(ke_serial.items(), key=lambda kv: kv[1]["ms"])
            ach = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] > 0 else 0.0
            traffic, traffic_src = pmc_traffic(name, config_tag(args))
            out["roofline"] = {"bound": "mfma", "kernel": name, "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                               "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
                               "launches": d["n"], "avg_launch_us": round(d["ms"] * 1000.0 / max(d["n"], 1), 2),
                               "instrumented_steps": serial_steps,
                               "how": "HIP events around every launch of the class in %d eager steps after the timed region, weight-gradient "
                                      "side stream off (one kernel on the GPU at a time; comparable with a serial rocprofv3 trace)" % serial_steps}
            do = ke.get(name)
            if do and do["ms"] > 0:
                ach_o = do["flops"] / (do["ms"] * 1e-3) / 1e12
                out["roofline"]["overlapped"] = {"achieved": round(ach_o, 2), "frac": round(ach_o / peak, 4),
                                                 "avg_launch_us": round(do["ms"] * 1000.0 / max(do["n"], 1), 2), "launches": do["n"],
                                                 "how": "same brackets with the production two-stream schedule: the launch time-shares "
                                                        "the GPU with the other stream, so it reads longer than in isolation"}
            out["kernel_classes_ms_per_step"] = {k: round(v["ms"] / serial_steps, 3) for k, v in sorted(ke_serial.items(), key=lambda kv: -kv[1]["ms"])}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
