"""batch_processor — the reference's ``training/batch_processor.py:10-59`` contract, restated.

(The reference file does not parse on Python >= 3.7: it passes ``async=False`` to ``.cuda()`` at
:20-21,24-25.)  Same signature and return structure:

    inputs, gts, saved_for_eval = batch_processor(state, batch)
    output, saved_for_loss = state.model(*inputs)
    loss, saved_for_log = model.build_loss(saved_for_loss, *gts)

``state.params.subnet_name`` selects the unpacking; ``state.params.gpus[0]`` is the target device.
With one process per GPU, each rank's loader yields its shard of the global batch (ddp.shard_batch).
"""
import torch


def _dev(state):
    gpus = getattr(state.params, "gpus", [0])
    return torch.device("cuda", gpus[0])


def batch_processor(state, batch):
    subnet_name = state.params.subnet_name      # 'detection_subnet' / 'keypoint_subnet' / 'prn_subnet'
    dev = _dev(state)
    torch.cuda.set_device(dev)         # kernels go to the current device's stream (the reference relies on device 0 being current)
    grad_ctx = torch.enable_grad() if state.model.training else torch.no_grad()
    with grad_ctx:
        if subnet_name == 'keypoint_subnet':
            inp, heat_temp, heat_weight = batch
            input_var = inp.to(dev, non_blocking=True)
            gts = [subnet_name, heat_temp.to(dev, non_blocking=True), heat_weight.to(dev, non_blocking=True)]
        elif subnet_name == 'detection_subnet':
            inp, anno = batch                   # anno: [x1, y1, x2, y2, category_id], padded with -1
            input_var = inp.to(dev, non_blocking=True)
            gts = [subnet_name, anno.to(dev, non_blocking=True)]
        elif subnet_name == 'train_both':       # SURVEY.md 8d combined step
            inp, heat_temp, heat_weight, anno = batch
            input_var = inp.to(dev, non_blocking=True)
            gts = [subnet_name, heat_temp.to(dev, non_blocking=True), heat_weight.to(dev, non_blocking=True),
                   anno.to(dev, non_blocking=True)]
        else:                                   # 'prn_subnet'
            inp, label = batch
            input_var = inp.to(dev, non_blocking=True).float()
            gts = [subnet_name, label.to(dev, non_blocking=True).float()]
    inputs = [[input_var, subnet_name]]
    saved_for_eval = []
    return inputs, gts, saved_for_eval


def train_step(model, optimizer, inputs, gts):
    """One iteration of Trainer._train_one_epoch's body (training/trainer.py:245-259)."""
    output, saved_for_loss = model(*inputs)
    loss, saved_for_log = model.build_loss(saved_for_loss, *gts)
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return loss, saved_for_log
