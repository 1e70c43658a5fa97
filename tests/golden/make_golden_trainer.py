#!/usr/bin/env python3
"""Fixture generator (build container only): drives the REAL reference ``training.trainer.Trainer`` and
``network.net_utils`` (imported from /root/reference) on the CPU with the stand-in network / loaders of tests/trainer_toy.py
and records what they DO — learning rate seen by every epoch, files present after every epoch (periodic / step / ``.best``
checkpoints, optimizer pickles, pruning), BatchNorm and module modes during training, validation batches consumed, resume
behaviour of every TrainParams switch, and net_utils' load/save rules — into tests/golden/g14_trainer.json.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_trainer.py

Shims (the reference targets torch 0.4 / CUDA / h5py; none of them alters the trainer's logic):
  * ``h5py``: absent here -> a minimal File / create_dataset / attrs object backed by one .npz per file (this fixture records
    WHICH files exist and what loads back, not HDF5 bytes; the byte format is pinned separately by g11_*.h5 from the real h5py);
  * ``.cuda()`` on tensors / modules: identity (CPU-only container);
  * ``np.float``: removed in numpy >= 1.24 (trainer.py:25, net_utils.py:98) -> float;
  * ``_LRScheduler``: in torch 0.4 ReduceLROnPlateau is NOT an _LRScheduler, in torch 2 it is; the isinstance tests at
    trainer.py:140,186 are given the 0.4 meaning (an _LRScheduler that is not a ReduceLROnPlateau).
"""
import json
import logging
import os
import shutil
import sys
import tempfile
import types
import warnings

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")
warnings.filterwarnings("ignore")

# ------------------------------------------------------------------------------------------------ shims
np.float = float


class _Attrs(dict):
    pass


class _FakeH5File(object):
    def __init__(self, fname, mode='r'):
        self.fname, self.mode = fname, mode
        self.data, self.attrs = {}, _Attrs()
        if mode == 'r':
            with open(fname, 'rb') as f:
                z = np.load(f, allow_pickle=True)
                for k in z.files:
                    if k.startswith('__attr__'):
                        self.attrs[k[8:]] = z[k].item() if z[k].ndim == 0 else z[k]
                    else:
                        self.data[k] = z[k]

    def __enter__(self):
        return self

    def __exit__(self, *a):
        if self.mode == 'w':
            out = dict(self.data)
            for k, v in self.attrs.items():
                out['__attr__' + k] = np.asarray(v)
            with open(self.fname, 'wb') as f:
                np.savez(f, **out)

    def create_dataset(self, k, data=None):
        self.data[k] = np.asarray(data)

    def keys(self):
        return self.data.keys()

    def __contains__(self, k):
        return k in self.data

    def __getitem__(self, k):
        return self.data[k]


sys.modules['h5py'] = types.SimpleNamespace(File=_FakeH5File)
torch.Tensor.cuda = lambda self, *a, **k: self
nn.Module.cuda = lambda self, *a, **k: self

import training.trainer as ref_trainer          # noqa: E402  (the real reference)
import network.net_utils as ref_net_utils       # noqa: E402
from torch.optim.lr_scheduler import LRScheduler, ReduceLROnPlateau, StepLR      # noqa: E402


class _Torch04Meta(type):
    def __instancecheck__(cls, obj):
        return isinstance(obj, LRScheduler) and not isinstance(obj, ReduceLROnPlateau)


class _Torch04LRScheduler(metaclass=_Torch04Meta):
    pass


ref_trainer._LRScheduler = _Torch04LRScheduler

from trainer_toy import ScriptedLoader, ToyNet, toy_batch_processor      # noqa: E402


class _Capture(logging.Handler):
    def __init__(self):
        logging.Handler.__init__(self)
        self.lines = []

    def emit(self, record):
        self.lines.append((record.levelname, record.getMessage()))


CAP = _Capture()
logging.getLogger('root').addHandler(CAP)
for h in list(logging.getLogger('root').handlers):
    if h is not CAP:
        logging.getLogger('root').removeHandler(h)


def listing(d):
    return sorted(os.listdir(d))


def bn_flags(model):
    m = model.module if hasattr(model, 'module') else model
    return [bool(x.training) for x in m.modules() if isinstance(x, nn.BatchNorm2d)]


def run_trainer(save_dir, subnet, max_epoch, train_n, train_vals, val_n, val_vals, sched, opts, lr=1e-2, run=True):
    """One real Trainer life: construct (-> resume facts), optionally train (-> per-epoch facts)."""
    torch.manual_seed(0)
    model = ToyNet()
    P = ref_trainer.TrainParams()
    P.exp_name, P.subnet_name, P.batch_size, P.max_epoch = 'toy', subnet, 2, max_epoch
    P.save_dir = save_dir
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
    env_before = os.environ.get('CUDA_VISIBLE_DEVICES')
    tr = ref_trainer.Trainer(model, P, toy_batch_processor, train_data, val_data)
    out = {'after_ctor': {'last_epoch': int(tr.last_epoch), 'lr': [float(g['lr']) for g in tr.optimizer.param_groups],
                          'adam_steps': sorted({int(float(s['step'])) for s in tr.optimizer.state.values()}),
                          'model_training': bool(tr.model.training), 'bn_training': bn_flags(tr.model),
                          'weight0': float(model.conv.weight.flatten()[0])}}
    if run:
        tr._toy_record = []
        snaps = []
        tr.on_start_epoch_hooks = [lambda t: snaps.append({'before_epoch': int(t.last_epoch), 'files': listing(save_dir)})]
        tr.train()
        out['epochs_seen_by_batches'] = tr._toy_record
        out['files_before_each_epoch'] = snaps
        out['files_at_end'] = listing(save_dir)
        out['last_epoch_at_end'] = int(tr.last_epoch)
        out['lr_at_end'] = [float(g['lr']) for g in tr.optimizer.param_groups]
        out['model_training_at_end'] = bool(tr.model.training)
        out['bn_training_at_end'] = bn_flags(tr.model)
        out['train_batches_served'] = train_data.served
        out['val_batches_served'] = val_data.served if val_data is not None else []
        out['log_value_kinds'] = {k: type(v).__name__ for k, v in tr.log_values.items()}
    if env_before is None:
        os.environ.pop('CUDA_VISIBLE_DEVICES', None)
    else:
        os.environ['CUDA_VISIBLE_DEVICES'] = env_before
    return out


def trainer_scenarios(root):
    res = {}
    val_vals = [1.03125, 0.78125, 0.90625, 0.65625, 0.71875, 0.59375]
    d1 = os.path.join(root, 's1')
    res['plateau_keypoint'] = {'args': dict(subnet='keypoint_subnet', max_epoch=5, train_n=3, train_vals=[2.0], val_n=4, val_vals=val_vals,
                                            sched='plateau', opts={'save_nckpt_max': 2, 'val_nbatch_end_epoch': 1})}
    res['plateau_keypoint']['out'] = run_trainer(d1, **res['plateau_keypoint']['args'])
    # resume variants on copies of that directory
    for name, opts, max_epoch, run in (('resume_default', {}, 6, True), ('resume_zero_epoch', {'zero_epoch': True}, 5, False),
                                       ('resume_ignore_opt_state', {'ignore_opt_state': True}, 5, False),
                                       ('resume_re_init', {'re_init': True}, 5, False),
                                       ('resume_explicit_ckpt', {'ckpt': 'ckpt_4.h5'}, 5, False)):
        d = os.path.join(root, name)
        shutil.copytree(d1, d)
        o = dict(opts)
        if 'ckpt' in o:
            o['ckpt'] = os.path.join(d, o['ckpt'])
        args = dict(subnet='keypoint_subnet', max_epoch=max_epoch, train_n=3, train_vals=[2.0], val_n=4, val_vals=val_vals[5:] + val_vals,
                    sched='plateau', opts=dict(o, save_nckpt_max=2, val_nbatch_end_epoch=1))
        out = run_trainer(d, run=run, **args)
        args['opts'] = dict(opts, save_nckpt_max=2, val_nbatch_end_epoch=1)        # paths are rebuilt by the test
        res[name] = {'args': args, 'out': out, 'run': run}
    res['steplr_detection'] = {'args': dict(subnet='detection_subnet', max_epoch=4, train_n=2, train_vals=[1.5], val_n=0, val_vals=[],
                                            sched='step', opts={'save_freq_epoch': 2, 'val_nbatch_end_epoch': 0})}
    res['steplr_detection']['out'] = run_trainer(os.path.join(root, 's2'), **res['steplr_detection']['args'])
    res['detection_with_validation'] = {'args': dict(subnet='detection_subnet', max_epoch=3, train_n=2, train_vals=[1.5], val_n=2,
                                                     val_vals=[0.5, 0.75, 0.25], sched=None, opts={'val_nbatch_end_epoch': 5})}
    res['detection_with_validation']['out'] = run_trainer(os.path.join(root, 's3'), **res['detection_with_validation']['args'])
    res['step_checkpoints'] = {'args': dict(subnet='keypoint_subnet', max_epoch=2, train_n=5, train_vals=[1.0], val_n=0, val_vals=[],
                                            sched=None, opts={'save_freq_step': 2, 'val_nbatch_end_epoch': 0})}
    res['step_checkpoints']['out'] = run_trainer(os.path.join(root, 's4'), **res['step_checkpoints']['args'])
    res['sparse_epoch_checkpoints'] = {'args': dict(subnet='keypoint_subnet', max_epoch=3, train_n=2, train_vals=[1.0], val_n=2, val_vals=[0.5, 0.4, 0.3],
                                                    sched=None, opts={'save_freq_epoch': 3, 'val_nbatch_end_epoch': 1})}
    res['sparse_epoch_checkpoints']['out'] = run_trainer(os.path.join(root, 's5'), **res['sparse_epoch_checkpoints']['args'])
    return res


def params_contract():
    P = ref_trainer.TrainParams()
    sd = P.state_dict()
    out = {'fields': list(sd.keys()), 'defaults': {k: (v if isinstance(v, (int, float, str, bool, type(None))) else repr(v)) for k, v in sd.items()}}
    CAP.lines.clear()
    P.update({'max_epoch': 7, 'no_such_option': 1})
    out['update_unknown_warns'] = [m for lv, m in CAP.lines if lv == 'WARNING']
    out['max_epoch_after_update'] = P.max_epoch
    out['str_head'] = str(P).splitlines()[0]
    return out


def net_utils_cases(root):
    """save_net / load_net rules (net_utils.py:30-110) on the stand-in network."""
    out = {}
    d = os.path.join(root, 'nu')
    os.makedirs(d)
    torch.manual_seed(1)
    net = ToyNet()
    opt = torch.optim.Adam(net.parameters(), lr=3e-3)
    y, saved = net([torch.randn(2, 1, 4, 4), 'x'])
    loss, _ = ToyNet.build_loss(saved, 'x', torch.ones(2, 3))
    loss.backward()
    opt.step()
    # pruning: numeric order of the trailing index, optimizer pickles of other checkpoints removed
    for e in (2, 9, 10):
        ref_net_utils.save_net(os.path.join(d, 'ckpt_%d.h5' % e), net, epoch=e, optimizers=[opt], rm_prev_opt=True, max_n_ckpts=2)
    out['files_after_three_saves_keep2'] = listing(d)
    ref_net_utils.save_net(os.path.join(d, 'ckpt_11.h5'), net, epoch=11)          # no optimizers: no pickle, no pruning at all
    out['files_after_save_without_optimizer'] = listing(d)
    ref_net_utils.save_net(os.path.join(d, 'ckpt_12.h5'), net, epoch=12, optimizers=[opt], rm_prev_opt=False, max_n_ckpts=-1)
    out['files_after_save_keep_all'] = listing(d)
    # load: (epoch, lr) form and (epoch, state_dicts) form
    fresh = ToyNet()
    with torch.no_grad():
        for p in fresh.parameters():
            p.add_(1.0)
    e, lr = ref_net_utils.load_net(os.path.join(d, 'ckpt_10.h5'), fresh)
    out['load_plain'] = {'epoch': int(e), 'lr': np.asarray(lr).tolist(), 'equal': bool(all(torch.equal(a, b) for a, b in zip(fresh.state_dict().values(), net.state_dict().values())))}
    e, sds = ref_net_utils.load_net(os.path.join(d, 'ckpt_10.h5'), ToyNet(), load_state_dict=True)
    out['load_with_state'] = {'epoch': int(e), 'n_state_dicts': len(sds), 'lr': float(sds[0]['param_groups'][0]['lr']),
                              'state_on_cpu': bool(all(v['exp_avg'].device.type == 'cpu' for v in sds[0]['state'].values())),
                              'adam_step': sorted({int(float(v['step'])) for v in sds[0]['state'].values()})}
    e, sds = ref_net_utils.load_net(os.path.join(d, 'ckpt_11.h5'), ToyNet(), load_state_dict=True)
    out['load_with_state_missing_pickle'] = {'epoch': int(e), 'state_dicts_is_none': sds is None}
    # a file saved from nn.DataParallel ('module.' prefix on every name) loads into a bare network
    with _FakeH5File(os.path.join(d, 'mod_1.h5'), 'w') as f:
        for k, v in net.state_dict().items():
            f.create_dataset('module.' + k, data=v.numpy())
    fresh = ToyNet()
    with torch.no_grad():
        fresh.conv.weight.add_(1.0)
    CAP.lines.clear()
    e, lr = ref_net_utils.load_net(os.path.join(d, 'mod_1.h5'), fresh)
    out['load_module_prefixed'] = {'epoch': int(e), 'equal': bool(torch.equal(fresh.conv.weight, net.conv.weight)),
                                   'warnings': len([1 for lv, m in CAP.lines if lv == 'WARNING'])}
    # missing layer + inconsistent shape: warnings, the rest still loads
    with _FakeH5File(os.path.join(d, 'odd_3.h5'), 'w') as f:
        for k, v in net.state_dict().items():
            if k == 'bn.bias':
                continue
            f.create_dataset(k, data=(np.zeros((3, 1, 1, 1), np.float32) if k == 'conv.weight' else v.numpy()))
        f.attrs['epoch'] = 3
        f.attrs['lr'] = 0.125
    fresh = ToyNet()
    before = fresh.conv.weight.clone()
    CAP.lines.clear()
    e, lr = ref_net_utils.load_net(os.path.join(d, 'odd_3.h5'), fresh)
    out['load_odd'] = {'epoch': int(e), 'lr': np.asarray(lr).tolist(), 'conv_untouched': bool(torch.equal(fresh.conv.weight, before)),
                       'bn_weight_loaded': bool(torch.equal(fresh.bn.weight, net.bn.weight)),
                       'warnings': sorted(m.split(':')[0] for lv, m in CAP.lines if lv == 'WARNING')}
    # set_optimizer_state_devices(state, None): tensors to the CPU, non-tensors untouched
    st = {0: {'step': 3, 'exp_avg': torch.ones(2)}}
    st2 = ref_net_utils.set_optimizer_state_devices(st, None)
    out['set_devices'] = {'same_object': st2 is st, 'step': st2[0]['step'], 'device': st2[0]['exp_avg'].device.type}
    return out


def main():
    root = tempfile.mkdtemp(prefix='mpn_ref_trainer_')
    try:
        fixture = {'generator': 'tests/golden/make_golden_trainer.py (real /root/reference training.trainer + network.net_utils, torch %s)' % torch.__version__,
                   'params': params_contract(), 'trainer': trainer_scenarios(root), 'net_utils': net_utils_cases(root)}
    finally:
        shutil.rmtree(root, ignore_errors=True)
    with open(os.path.join(HERE, 'g14_trainer.json'), 'w') as f:
        json.dump(fixture, f, indent=1, sort_keys=True)
    print('wrote g14_trainer.json')


if __name__ == '__main__':
    main()
