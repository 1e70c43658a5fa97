"""Worker of tests/test_round3_cpu.py::test_trainer_under_two_ranks_writes_once_and_decides_together (gloo, CPU, stand-in network)."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from torch.optim.lr_scheduler import ReduceLROnPlateau
    from multiposenet.pytorch_amd.training.trainer import Trainer, TrainParams
    from trainer_toy import ScriptedLoader, ToyNet, toy_batch_processor
    torch.manual_seed(0)
    model = ToyNet()
    P = TrainParams(exp_name='toy2', subnet_name='keypoint_subnet', batch_size=2, max_epoch=3, gpus=[], save_dir=os.environ["MPN_TRAINER_DIR"],
                    save_nckpt_max=2, val_nbatch_end_epoch=1, print_freq=1)
    P.optimizer = torch.optim.Adam(model.parameters(), lr=1e-2)
    P.lr_scheduler = ReduceLROnPlateau(P.optimizer, mode='min', factor=0.5, patience=0, threshold=0.0)
    # local validation losses: rank 0 sees 1.0, 0.5, 0.25 (always improving), rank 1 sees 0.5, 1.5, 0.25 -> means 0.75, 1.0, 0.25
    vals = ([1.0, 0.5, 0.25], [0.5, 1.5, 0.25])[rank]
    tr = Trainer(model, P, toy_batch_processor, ScriptedLoader(2, [2.0], seed=1), ScriptedLoader(2, vals, seed=2))
    tr.train()
    out = {"lr": [float(g['lr']) for g in P.optimizer.param_groups], "last_epoch": tr.last_epoch,
           "files": sorted(os.listdir(P.save_dir))}
    dist.barrier()
    with open(os.path.join(os.environ["MPN_TRAINER_OUT"], "rank%d.json" % rank), "w") as f:
        json.dump(out, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
