"""Worker of tests/test_round2_gpu.py::test_two_rank_data_parallel_equals_global_batch_on_device (one process per rank).

Each rank runs the HIP forward/backward on ITS shard of a fixed global batch (fp32 kernels, BatchNorm frozen so that
images are independent), the GradReducer averages the gradient arena across the two ranks while backward is still
enqueueing, and rank 0 additionally computes the global-batch gradient alone for comparison."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    ngpu = torch.cuda.device_count()
    backend = "nccl" if ngpu >= world else "gloo"
    local = rank if ngpu >= world else 0
    torch.cuda.set_device(local)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from multiposenet.pytorch_amd import ddp
    from multiposenet.pytorch_amd.network.posenet import poseNet
    from multiposenet.pytorch_amd import synthetic as weightgen

    torch.manual_seed(100 + rank)           # ranks start different: attach() must broadcast rank 0's parameters
    m = poseNet(50, compute_dtype=torch.float32).cuda()
    if rank == 0:
        shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        sd = weightgen.gen_state_dict(shapes, seed=0, flavour="he", skip_prefixes=("prn.",))
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    for p in m.prn.parameters():
        p.requires_grad = False
    m.train()
    m.freeze_bn()                            # frozen BN: per-image independence, shard mean == global mean exactly
    B, S = 4, 96
    img = torch.from_numpy(weightgen.gen_images(7, B, S, S)).cuda()
    heat, wgt = (torch.from_numpy(a).cuda() for a in weightgen.gen_keypoint_gt(8, B, S // 4, S // 4))
    anno = torch.from_numpy(weightgen.gen_boxes_gt(9, B, S)).cuda()

    def step(model, sl):
        model._arena.ensure_grads()
        model._arena.grad_flat.zero_()
        pred, saved = model([img[sl].contiguous(), "train_both"])
        loss, _ = poseNet.build_loss(saved, "train_both", heat[sl].contiguous(), wgt[sl].contiguous(), anno[sl].contiguous())
        loss.backward()
        torch.cuda.synchronize()
        return float(loss)

    k = B // world
    if os.environ.get("MPN_DDP_MODE") == "replay":
        # three optimizer steps of the data-parallel job, once through the eager step and once through the recorded launch list
        # (collectives are part of the list): parameters must agree bit for bit, on every rank
        import copy
        from multiposenet.pytorch_amd.optim import FusedAdam
        from multiposenet.pytorch_amd.replay import ReplayedTrainStep
        from multiposenet.pytorch_amd.training.batch_processor import train_step
        sl = slice(rank * k, (rank + 1) * k)
        inputs = [[img[sl].contiguous(), "train_both"]]
        gts = ["train_both", heat[sl].contiguous(), wgt[sl].contiguous(), anno[sl].contiguous()]
        ddp.attach(m, bucket_mb=8.0)
        start = copy.deepcopy(m.state_dict())
        res = {}
        for mode in ("eager", "replay"):
            m.load_state_dict(start)
            opt = FusedAdam(m, lr=1e-3, weight_decay=0.0)
            stepper = ReplayedTrainStep(m, opt) if mode == "replay" else None
            losses = []
            for _ in range(4):
                loss, _ = stepper(inputs, gts) if stepper is not None else train_step(m, opt, inputs, gts)
                losses.append(float(loss))
            torch.cuda.synchronize()
            res[mode + "_params"] = m._arena.flat.detach().cpu().numpy().copy()
            res[mode + "_loss"] = np.array(losses)
            if stepper is not None:
                res["replays"] = stepper.replays
                # round 6: the recorded data-parallel step updates bucket by bucket behind each all-reduce (finishing stream)
                res["bucket_updates"] = m._reducer.updated
                res["buckets"] = len(m._reducer.buckets)
                res["bucketed"] = int(stepper.bucketed_update)
        np.savez(os.path.join(os.environ["MPN_DDP_OUT"], "rank%d.npz" % rank), backend=backend, **res)
        dist.barrier()
        dist.destroy_process_group()
        return
    red = ddp.attach(m, bucket_mb=8.0)
    loss = step(m, slice(rank * k, (rank + 1) * k))
    grad = m._arena.grad_flat.cpu().numpy()
    lt = torch.tensor([loss], dtype=torch.float64)
    if backend == "nccl":
        lt = lt.cuda()
    dist.all_reduce(lt)
    out = {"grad": grad, "buckets": np.array(red.signature(), dtype=np.int64), "launched": red.launched,
           "backend": backend, "loss_mean": float(lt.item()) / world}
    if rank == 0:
        m._reducer = None
        out["global_loss"] = step(m, slice(0, B))
        out["global_grad"] = m._arena.grad_flat.cpu().numpy()
    np.savez(os.path.join(os.environ["MPN_DDP_OUT"], "rank%d.npz" % rank), **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
