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


# ------------------------------------------------------------------------------------------------ Trainer / net_utils vs the real reference
def _fixture():
    with open(os.path.join(ROOT, "tests", "golden", "g14_trainer.json")) as f:
        return json.load(f)


def _bn_flags(model):
    return [bool(m.training) for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)]


def _drive_trainer(save_dir, subnet, max_epoch, train_n, train_vals, val_n, val_vals, sched, opts, lr=1e-2, run=True, device=None):
    """The mirror image of tests/golden/make_golden_trainer.py:run_trainer for THIS repository's Trainer."""
    from torch.optim.lr_scheduler import ReduceLROnPlateau, StepLR
    from multiposenet.pytorch_amd.training.trainer import Trainer, TrainParams
    from trainer_toy import ScriptedLoader, ToyNet, toy_batch_processor
    torch.manual_seed(0)
    model = ToyNet()
    P = TrainParams()
    P.exp_name, P.subnet_name, P.batch_size, P.max_epoch = 'toy', subnet, 2, max_epoch
    P.save_dir = save_dir
    P.gpus = [] if device is None else [device]
    if device is not None:
        model = model.to(torch.device('cuda', device))
    P.optimizer = torch.optim.Adam(model.parameters(), lr=lr)
    if sched == 'plateau':
        P.lr_scheduler = ReduceLROnPlateau(P.optimizer, mode='min', factor=0.5, patience=0, threshold=0.0)
    elif sched == 'step':
        P.lr_scheduler = StepLR(P.optimizer, step_size=1, gamma=0.1)
    P.print_freq = 2
    for k, v in opts.items():
        setattr(P, k, v)
    train_data = ScriptedLoader(train_n, train_vals, seed=1)
    val_data = ScriptedLoader(val_n, val_vals, seed=2) if val_n else None
    tr = Trainer(model, P, toy_batch_processor, train_data, val_data)
    out = {'after_ctor': {'last_epoch': int(tr.last_epoch), 'lr': [float(g['lr']) for g in tr.optimizer.param_groups],
                          'adam_steps': sorted({int(float(s['step'])) for s in tr.optimizer.state.values()}),
                          'model_training': bool(tr.model.training), 'bn_training': _bn_flags(tr.model),
                          'weight0': float(model.conv.weight.flatten()[0])}}
    if run:
        tr._toy_record = []
        snaps = []
        tr.on_start_epoch_hooks = [lambda t: snaps.append({'before_epoch': int(t.last_epoch), 'files': sorted(os.listdir(save_dir))})]
        tr.train()
        out['epochs_seen_by_batches'] = tr._toy_record
        out['files_before_each_epoch'] = snaps
        out['files_at_end'] = sorted(os.listdir(save_dir))
        out['last_epoch_at_end'] = int(tr.last_epoch)
        out['lr_at_end'] = [float(g['lr']) for g in tr.optimizer.param_groups]
        out['model_training_at_end'] = bool(tr.model.training)
        out['bn_training_at_end'] = _bn_flags(tr.model)
        out['train_batches_served'] = train_data.served
        out['val_batches_served'] = val_data.served if val_data is not None else []
        out['log_value_kinds'] = {k: ('number' if hasattr(v, 'value') else type(v).__name__) for k, v in tr.log_values.items()}
        if device is not None:
            assert all(s['exp_avg'].is_cuda for s in tr.optimizer.state.values())
    return out


def check_trainer_against_reference(tmp_path, device=None):
    """Every scenario of the fixture through this repository's Trainer: identical learning rates per batch, module / BatchNorm
    modes per batch, files before every epoch and at the end (names carry the validation loss to 5 decimals), batches consumed,
    resume facts for every TrainParams switch."""
    import shutil
    fx = _fixture()['trainer']
    root = str(tmp_path)
    made = {}
    order = ['plateau_keypoint', 'resume_default', 'resume_zero_epoch', 'resume_ignore_opt_state', 'resume_re_init', 'resume_explicit_ckpt',
             'steplr_detection', 'detection_with_validation', 'step_checkpoints', 'sparse_epoch_checkpoints']
    assert sorted(order) == sorted(fx.keys())
    for name in order:
        sc = fx[name]
        args = dict(sc['args'])
        opts = dict(args.pop('opts'))
        d = os.path.join(root, name)
        if name.startswith('resume_'):
            shutil.copytree(made['plateau_keypoint'], d)
            if 'ckpt' in opts:
                opts['ckpt'] = os.path.join(d, opts['ckpt'])
        made[name] = d
        got = _drive_trainer(d, opts=opts, run=sc.get('run', True), device=device, **args)
        want = sc['out']
        for key in want:
            w, g = want[key], got[key]
            if key == 'after_ctor':
                assert abs(g.pop('weight0') - w['weight0']) <= 1e-6, (name, 'weight0')
                w = {k: v for k, v in w.items() if k != 'weight0'}
            if key == 'log_value_kinds':
                w = {k: ('number' if v == 'AverageValueMeter' else v) for k, v in w.items()}
            assert g == w, "scenario %s: %s differs from the reference Trainer\n got  %s\n want %s" % (name, key, g, w)
    return len(order)


def test_trainer_follows_the_real_reference_trainer_on_cpu(tmp_path):
    n = check_trainer_against_reference(tmp_path)
    assert n == 10


def test_train_params_contract():
    from multiposenet.pytorch_amd.training.trainer import TrainParams
    fx = _fixture()['params']
    P = TrainParams()
    sd = P.state_dict()
    assert list(sd.keys()) == fx['fields']
    for k, v in fx['defaults'].items():
        mine = sd[k]
        assert (mine if isinstance(mine, (int, float, str, bool, type(None))) else repr(mine)) == v, k
    import logging
    seen = []
    h = logging.Handler()
    h.emit = lambda rec: seen.append(rec.getMessage())
    logging.getLogger("multiposenet").addHandler(h)
    try:
        P.update({'max_epoch': 7, 'no_such_option': 1})
    finally:
        logging.getLogger("multiposenet").removeHandler(h)
    assert seen == fx['update_unknown_warns'] and P.max_epoch == fx['max_epoch_after_update'] and not hasattr(P, 'no_such_option')
    assert str(P).splitlines()[0] == fx['str_head']
    assert TrainParams(max_epoch=3).max_epoch == 3 and TrainParams().gpus is not TrainParams().gpus


def test_net_utils_follow_the_real_reference_functions(tmp_path):
    """save_net / load_net / set_optimizer_state_devices on the stand-in network: the same files survive pruning, the same
    values come back, the same things only warn — as recorded from the reference's own functions."""
    import logging
    from multiposenet.pytorch_amd.network import hdf5min, net_utils
    from trainer_toy import ToyNet
    fx = _fixture()['net_utils']
    d = str(tmp_path)
    torch.manual_seed(1)
    net = ToyNet()
    opt = torch.optim.Adam(net.parameters(), lr=3e-3)
    y, saved = net([torch.randn(2, 1, 4, 4), 'x'])
    loss, _ = ToyNet.build_loss(saved, 'x', torch.ones(2, 3))
    loss.backward()
    opt.step()
    for e in (2, 9, 10):
        net_utils.save_net(os.path.join(d, 'ckpt_%d.h5' % e), net, epoch=e, optimizers=[opt], rm_prev_opt=True, max_n_ckpts=2)
    assert sorted(os.listdir(d)) == fx['files_after_three_saves_keep2']
    net_utils.save_net(os.path.join(d, 'ckpt_11.h5'), net, epoch=11)
    assert sorted(os.listdir(d)) == fx['files_after_save_without_optimizer']
    net_utils.save_net(os.path.join(d, 'ckpt_12.h5'), net, epoch=12, optimizers=[opt], rm_prev_opt=False, max_n_ckpts=-1)
    assert sorted(os.listdir(d)) == fx['files_after_save_keep_all']
    fresh = ToyNet()
    with torch.no_grad():
        for p in fresh.parameters():
            p.add_(1.0)
    e, lr = net_utils.load_net(os.path.join(d, 'ckpt_10.h5'), fresh)
    assert {'epoch': int(e), 'lr': np.asarray(lr).tolist(),
            'equal': all(torch.equal(a, b) for a, b in zip(fresh.state_dict().values(), net.state_dict().values()))} == fx['load_plain']
    e, sds = net_utils.load_net(os.path.join(d, 'ckpt_10.h5'), ToyNet(), load_state_dict=True)
    assert {'epoch': int(e), 'n_state_dicts': len(sds), 'lr': float(sds[0]['param_groups'][0]['lr']),
            'state_on_cpu': all(v['exp_avg'].device.type == 'cpu' for v in sds[0]['state'].values()),
            'adam_step': sorted({int(float(v['step'])) for v in sds[0]['state'].values()})} == fx['load_with_state']
    e, sds = net_utils.load_net(os.path.join(d, 'ckpt_11.h5'), ToyNet(), load_state_dict=True)
    assert {'epoch': int(e), 'state_dicts_is_none': sds is None} == fx['load_with_state_missing_pickle']
    seen = []
    h = logging.Handler()
    h.emit = lambda rec: seen.append((rec.levelname, rec.getMessage()))
    logging.getLogger("multiposenet").addHandler(h)
    try:
        hdf5min.write_file(os.path.join(d, 'mod_1.h5'), [('module.' + k, v.numpy()) for k, v in net.state_dict().items()])
        fresh = ToyNet()
        with torch.no_grad():
            fresh.conv.weight.add_(1.0)
        e, lr = net_utils.load_net(os.path.join(d, 'mod_1.h5'), fresh)
        assert {'epoch': int(e), 'equal': bool(torch.equal(fresh.conv.weight, net.conv.weight)),
                'warnings': len([1 for lv, m in seen if lv == 'WARNING'])} == fx['load_module_prefixed']
        arrays = [(k, (np.zeros((3, 1, 1, 1), np.float32) if k == 'conv.weight' else v.numpy())) for k, v in net.state_dict().items() if k != 'bn.bias']
        hdf5min.write_file(os.path.join(d, 'odd_3.h5'), arrays, attrs={'epoch': np.int64(3), 'lr': np.float64(0.125)})
        fresh = ToyNet()
        before = fresh.conv.weight.clone()
        del seen[:]
        e, lr = net_utils.load_net(os.path.join(d, 'odd_3.h5'), fresh)
        assert {'epoch': int(e), 'lr': np.asarray(lr).tolist(), 'conv_untouched': bool(torch.equal(fresh.conv.weight, before)),
                'bn_weight_loaded': bool(torch.equal(fresh.bn.weight, net.bn.weight)),
                'warnings': sorted(m.split(':')[0] for lv, m in seen if lv == 'WARNING')} == fx['load_odd']
    finally:
        logging.getLogger("multiposenet").removeHandler(h)
    st = {0: {'step': 3, 'exp_avg': torch.ones(2)}}
    st2 = net_utils.set_optimizer_state_devices(st, None)
    assert {'same_object': st2 is st, 'step': st2[0]['step'], 'device': st2[0]['exp_avg'].device.type} == fx['set_devices']
    assert not [n for n in os.listdir(d) if n.endswith('~')], "temporary files left behind"


def test_trainer_under_two_ranks_writes_once_and_decides_together(tmp_path):
    """under torch.distributed every rank used to save / prune / copy checkpoints in the same directory and
    to step ReduceLROnPlateau on its LOCAL validation loss.  Two gloo ranks whose local validation losses disagree about what
    improved: only rank 0 writes (no crash on files another rank removed), the best-checkpoint copies and the learning rate
    follow the MEAN validation loss, both ranks end in the same state."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    run_dir, out_dir = str(tmp_path / "run"), str(tmp_path / "out")
    os.makedirs(out_dir)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", MPN_TRAINER_DIR=run_dir, MPN_TRAINER_OUT=out_dir)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "trainer_ddp_worker.py")], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    logs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, lg) in enumerate(zip(procs, logs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, lg[-3000:])
    res = [json.load(open(os.path.join(out_dir, "rank%d.json" % r))) for r in range(2)]
    assert res[0] == res[1]
    files = res[0]["files"]
    best = sorted(f for f in files if f.endswith(".best"))
    # mean validation losses 0.75.., 1.00.., 0.25..: epochs 1 and 3 improve, epoch 2 does not (rank 0 alone would have said it did)
    assert [b.split("_")[1] for b in best] == ["1", "3"] and best[0].startswith("ckpt_1_0.75") and best[1].startswith("ckpt_3_0.25"), best
    assert [f for f in files if f.endswith(".h5")] == ["ckpt_2.h5", "ckpt_3.h5"] and [f for f in files if f.endswith(".pk")] == ["ckpt_3.h5.optimizer_state.pk"]
    assert res[0]["lr"] == [0.005] and res[0]["last_epoch"] == 3          # halved once, after epoch 2, on both ranks


def test_host_plans_of_the_round3_launch_options():
    """Host-side decisions that pick kernels: the in-launch BatchNorm finalize (up to 64 pixel tiles) and the geometry test of the
    one-pass heat-map loss."""
    import torch
    from multiposenet.pytorch_amd import ops
    from multiposenet.pytorch_amd.network import losses
    assert ops.fin_in_launch(1) and ops.fin_in_launch(ops.FIN_MAX_TILES) and not ops.fin_in_launch(ops.FIN_MAX_TILES + 1) and not ops.fin_in_launch(225)
    B, H, W = 2, 24, 40
    lv = [ops.Act(torch.zeros(B, H >> s, W >> s, 32), c) for s, c in ((0, 19), (1, 19), (2, 19), (3, 19), (0, 18))]
    heat = torch.zeros(B, 18, H, W)
    assert losses.mse_train_supported(lv, heat)
    assert not losses.mse_train_supported(lv, heat.permute(0, 1, 3, 2))                      # targets must be dense NCHW
    assert not losses.mse_train_supported(lv[:1] + [ops.Act(torch.zeros(B, 13, 20, 32), 19)] + lv[2:], heat)    # a level off the power-of-two grid
    assert not losses.mse_train_supported([ops.Act(a.t.half(), a.C) for a in lv], heat)      # internal predictions are f32
    assert not losses.mse_train_supported(lv, torch.zeros(B, 18, 20, 40))                    # sides must be multiples of 8
