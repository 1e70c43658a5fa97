"""Probe: can RCCL form a 2-rank communicator with both ranks on ONE GPU?  (VERDICT r2 item 1c asked for the two 2-rank device
tests to use RCCL on the single-GPU box "where the stack allows it".)  Prints the outcome; run under `timeout`."""
import os
import subprocess
import sys

if "RANK" not in os.environ:
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    ps = [subprocess.Popen([sys.executable, __file__], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    for r, p in enumerate(ps):
        try:
            out = p.communicate(timeout=90)[0]
        except subprocess.TimeoutExpired:
            p.kill()
            out = "TIMEOUT"
        print("rank %d rc=%s: %s" % (r, p.returncode, out[-600:].replace("\n", " | ")))
    sys.exit(0)
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", rank=int(os.environ["RANK"]), world_size=2, device_id=torch.device("cuda", 0))
    t = torch.ones(4, device="cuda") * (int(os.environ["RANK"]) + 1)
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print("RCCL_SHARED_GPU_OK", t.tolist())
except Exception as e:          # noqa: BLE001
    print("RCCL_SHARED_GPU_REFUSED", type(e).__name__, str(e)[:400])
