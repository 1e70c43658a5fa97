"""Checkpoint I/O with the reference's interface and file layout (``network/net_utils.py:12-110``):

    save_net(fname, net, epoch=-1, optimizers=None, rm_prev_opt=False, max_n_ckpts=-1)
    epoch, lr_or_state_dicts = load_net(fname, net, prefix='', load_state_dict=False)

A checkpoint is one flat HDF5 file — a dataset per ``net.state_dict()`` key (fp32 / int64 numpy arrays in the
reference's logical [Cout, Cin, R, S] layout, whatever the arena stores physically) plus the root attribute ``epoch`` —
and, when optimizers are given, ``fname + '.optimizer_state.pk'``: a pickle of ``[optimizer.state_dict(), ...]`` with
the tensors moved to the CPU.  h5py is used when it can be imported; otherwise the built-in ``hdf5min`` module reads
and writes the same bytes-on-disk subset, so the authors' ``ckpt_baseline_resnet101.h5`` loads either way.
"""
import logging
import os
import pickle
from copy import deepcopy

import numpy as np
import torch
import torch.nn as nn

from . import hdf5min

logger = logging.getLogger("multiposenet")

try:                                    # pragma: no cover  (not installed in the build image)
    import h5py as _h5py
except ImportError:
    _h5py = None


def set_optimizer_state_devices(state, device_id=None):
    """net_utils.py:12-27: move optimizer state tensors to the CPU (device_id None) or to cuda:device_id."""
    for k, v in state.items():
        for k2 in v.keys():
            if hasattr(v[k2], 'cuda'):
                if device_id is None:
                    v[k2] = v[k2].cpu()
                else:
                    v[k2] = v[k2].cuda(device_id)
    return state


def _write_h5(fname, arrays, epoch):
    if _h5py is not None:
        with _h5py.File(fname, mode='w') as h5f:
            for k, v in arrays:
                h5f.create_dataset(k, data=v)
            h5f.attrs['epoch'] = epoch
    else:
        hdf5min.write_file(fname, arrays, attrs={'epoch': np.int64(epoch)})


def _open_h5(fname):
    if _h5py is not None:
        return _h5py.File(fname, mode='r')
    return hdf5min.File(fname)


def save_net(fname, net, epoch=-1, optimizers=None, rm_prev_opt=False, max_n_ckpts=-1):
    """net_utils.py:30-66."""
    arrays = [(k, v.detach().cpu().contiguous().numpy()) for k, v in net.state_dict().items()]
    _write_h5(fname, arrays, epoch)

    if optimizers is not None:
        state_dicts = []
        for optimizer in optimizers:
            state_dict = deepcopy(optimizer.state_dict())
            state_dict['state'] = set_optimizer_state_devices(state_dict['state'], device_id=None)
            state_dicts.append(state_dict)
        state_file = fname + '.optimizer_state.pk'
        with open(state_file, 'wb') as f:
            pickle.dump(state_dicts, f)

        if rm_prev_opt:                                  # keep only the newest optimizer state (net_utils.py:49-55)
            root = os.path.split(fname)[0]
            for filename in os.listdir(root or '.'):
                filename = os.path.join(root, filename)
                if filename.endswith('.optimizer_state.pk') and filename != state_file:
                    logger.info('Remove {}'.format(filename))
                    os.remove(filename)

        if max_n_ckpts > 0:                              # prune old checkpoints by their trailing index (net_utils.py:58-66)
            root = os.path.split(fname)[0]
            ckpts = [f for f in os.listdir(root or '.') if os.path.splitext(f)[-1] == '.h5']
            ckpts = sorted(ckpts, key=lambda name: int(os.path.splitext(name)[0].split('_')[-1]))
            if len(ckpts) > max_n_ckpts:
                for ckpt in ckpts[0:-max_n_ckpts]:
                    filename = os.path.join(root, ckpt)
                    logger.info('Remove {}'.format(filename))
                    os.remove(filename)


def load_net(fname, net, prefix='', load_state_dict=False):
    """net_utils.py:69-110: copies every matching dataset into ``net.state_dict()`` in place; tolerates a 'module.'
    prefix on the stored names (files saved from nn.DataParallel), missing layers and shape mismatches (warnings)."""
    with _open_h5(fname) as h5f:
        h5f_is_module = True
        for k in h5f.keys():
            if not str(k).startswith('module.'):
                h5f_is_module = False
                break
        if prefix == '' and not isinstance(net, nn.DataParallel) and h5f_is_module:
            prefix = 'module.'

        for k, v in net.state_dict().items():
            k = prefix + k
            if k in h5f:
                param = torch.from_numpy(np.asarray(h5f[k]))
                if v.size() != param.size():
                    logger.warning('Inconsistent shape: {}, {}'.format(v.size(), param.size()))
                else:
                    v.copy_(param)
            else:
                logger.warning('No layer: {}'.format(k))

        epoch = h5f.attrs['epoch'] if 'epoch' in h5f.attrs else -1
        epoch = int(epoch)

        if not load_state_dict:
            if 'learning_rates' in h5f.attrs:
                lr = h5f.attrs['learning_rates']
            else:
                lr = h5f.attrs.get('lr', -1)
                lr = np.asarray([lr] if lr > 0 else [], dtype=float)          # np.float (net_utils.py:98) is gone in numpy 2
            return epoch, lr

    state_file = fname + '.optimizer_state.pk'
    if os.path.isfile(state_file):
        with open(state_file, 'rb') as f:
            state_dicts = pickle.load(f)
            if not isinstance(state_dicts, list):
                state_dicts = [state_dicts]
    else:
        state_dicts = None
    return epoch, state_dicts
