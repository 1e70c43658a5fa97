"""Checkpoint files in the reference's on-disk format, on top of the built-in HDF5 codec (``hdf5min``).

What a checkpoint IS is fixed by the reference (``network/net_utils.py:30-110``) and pinned by tests/golden/g14_trainer.json
(recorded from the real functions) and g11_*.h5 (written by the real h5py):

  ``<name>.h5``                      one little-endian dataset per ``state_dict()`` entry under its flat key, logical shapes
                                     ([Cout, Cin, R, S] for convolutions, whatever the arena stores), root attribute ``epoch``;
  ``<name>.h5.optimizer_state.pk``   pickle of ``[optimizer.state_dict(), ...]`` with every tensor on the CPU.

How it is done here is this stack's own: the parameter arena crosses PCIe as ONE transfer instead of one per tensor, files
appear atomically (written beside the target, then renamed), pruning tolerates another process having removed a file first,
and loading resolves names through an index of the file built once.  Entry points keep the reference's names / signatures:

    save_net(fname, net, epoch=-1, optimizers=None, rm_prev_opt=False, max_n_ckpts=-1)
    load_net(fname, net, prefix='', load_state_dict=False) -> (epoch, learning_rates | optimizer state dicts | None)
    set_optimizer_state_devices(state, device_id=None)
"""
import logging
import os
import pickle

import numpy as np
import torch
import torch.nn as nn

from . import hdf5min

logger = logging.getLogger("multiposenet")

OPT_SUFFIX = '.optimizer_state.pk'


# ---------------------------------------------------------------------------------------------------- small helpers
def _moved(obj, device):
    """Copy of a nested optimizer-state structure with every tensor on `device` (containers rebuilt, leaves shared)."""
    if torch.is_tensor(obj):
        return obj.detach().to(device)
    if isinstance(obj, dict):
        return type(obj)((k, _moved(v, device)) for k, v in obj.items())
    if isinstance(obj, (list, tuple)):
        return type(obj)(_moved(v, device) for v in obj)
    return obj


def set_optimizer_state_devices(state, device_id=None):
    """``optimizer.state`` (or the 'state' part of a state dict) moved IN PLACE: to the CPU when ``device_id`` is None, else
    to ``cuda:device_id``; entries that are not tensors stay as they are.  Returns ``state`` (net_utils.py:12-27)."""
    target = torch.device('cpu') if device_id is None else torch.device('cuda', int(device_id))
    for slot in state.values():
        for name in list(slot.keys()):
            if torch.is_tensor(slot[name]):
                slot[name] = slot[name].to(target)
    return state


def checkpoint_index(name):
    """Trailing integer of a checkpoint's stem ('ckpt_12.h5' -> 12): the order checkpoints are aged by."""
    stem = os.path.splitext(name)[0]
    return int(stem.split('_')[-1])


def list_checkpoints(folder):
    """``*.h5`` files of `folder`, oldest first.  ('.h5.best' / '.h5.ckpt' copies carry another extension and never count.)"""
    names = [n for n in os.listdir(folder or '.') if os.path.splitext(n)[-1] == '.h5']
    return sorted(names, key=checkpoint_index)


def _unlink(path):
    try:
        os.remove(path)
        logger.info('Remove {}'.format(path))
    except FileNotFoundError:            # another rank / process was faster
        pass


def _host_state(net):
    """[(key, numpy array)] of ``net.state_dict()``.  A network whose parameters live in a flat arena (poseNet) is fetched
    with a single device-to-host copy of the arena; the per-key arrays are host views in the logical layout."""
    sd = net.state_dict()
    arena = getattr(net, '_arena', None)
    host = {}
    if arena is not None and arena.flat.is_cuda and arena.consistent():
        flat = arena.flat.detach().cpu()
        by_ptr = {p.data_ptr(): i for i, p in enumerate(arena.params)}
        for k, v in sd.items():
            i = by_ptr.get(v.data_ptr())
            if i is not None and tuple(arena.params[i].shape) == tuple(v.shape):
                host[k] = arena._view(flat, i, v.shape)
    out = []
    for k, v in sd.items():
        t = host.get(k)
        if t is None:
            t = v.detach().cpu()
        a = t.numpy()
        out.append((k, a if a.flags.c_contiguous else a.copy(order='C')))      # (np.ascontiguousarray would turn 0-d into 1-d)
    return out


def _write_atomically(path, writer):
    tmp = '%s.tmp%d~' % (path, os.getpid())
    try:
        writer(tmp)
        os.replace(tmp, path)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)


# ---------------------------------------------------------------------------------------------------- save
def save_net(fname, net, epoch=-1, optimizers=None, rm_prev_opt=False, max_n_ckpts=-1):
    """Write ``fname`` (+ the optimizer pickle when ``optimizers`` is given).  Only with optimizers: ``rm_prev_opt`` deletes
    every OTHER ``*.optimizer_state.pk`` of the folder, ``max_n_ckpts > 0`` keeps the newest that many ``*.h5`` files."""
    arrays = _host_state(net)
    _write_atomically(fname, lambda p: hdf5min.write_file(p, arrays, attrs={'epoch': np.int64(epoch)}))
    if optimizers is None:
        return
    folder = os.path.split(fname)[0]
    state_file = fname + OPT_SUFFIX
    payload = [_moved(opt.state_dict(), torch.device('cpu')) for opt in optimizers]

    def dump(p):
        with open(p, 'wb') as f:
            pickle.dump(payload, f)
    _write_atomically(state_file, dump)
    if rm_prev_opt:
        for n in os.listdir(folder or '.'):
            p = os.path.join(folder, n)
            if n.endswith(OPT_SUFFIX) and p != state_file:
                _unlink(p)
    if max_n_ckpts > 0:
        names = list_checkpoints(folder)
        for n in names[:max(0, len(names) - max_n_ckpts)]:
            _unlink(os.path.join(folder, n))


# ---------------------------------------------------------------------------------------------------- load
def _is_data_parallel_file(names):
    """True when every stored name carries the 'module.' prefix nn.DataParallel adds."""
    return all(str(n).startswith('module.') for n in names)


def load_net(fname, net, prefix='', load_state_dict=False):
    """Copy the stored tensors into ``net.state_dict()`` in place.  A file saved from a DataParallel-wrapped model loads into
    a bare one; a name the file lacks or a tensor of another shape is reported with a warning and skipped.
    Returns ``(epoch, x)``: epoch is -1 when the file has no such attribute; x is the array of learning rates stored in the
    file's attributes (``learning_rates``, or ``[lr]`` when ``lr`` > 0, else empty) or, with ``load_state_dict=True``, the
    list unpickled from ``fname + '.optimizer_state.pk'`` (None if that file does not exist)."""
    with hdf5min.File(fname) as h5f:
        stored = set(h5f.keys())
        if prefix == '' and not isinstance(net, nn.DataParallel) and _is_data_parallel_file(stored):
            prefix = 'module.'
        with torch.no_grad():
            for key, dst in net.state_dict().items():
                name = prefix + key
                if name not in stored:
                    logger.warning('No layer: {}'.format(name))
                    continue
                src = torch.from_numpy(np.asarray(h5f[name]))
                if tuple(src.shape) != tuple(dst.shape):
                    logger.warning('Inconsistent shape: {}, {}'.format(tuple(dst.shape), tuple(src.shape)))
                    continue
                dst.copy_(src)
        attrs = h5f.attrs
        epoch = int(attrs['epoch']) if 'epoch' in attrs else -1
        if not load_state_dict:
            if 'learning_rates' in attrs:
                return epoch, np.asarray(attrs['learning_rates'], dtype=float)
            lr = float(attrs.get('lr', -1))
            return epoch, np.asarray([lr] if lr > 0 else [], dtype=float)
    state_file = fname + OPT_SUFFIX
    if not os.path.isfile(state_file):
        return epoch, None
    with open(state_file, 'rb') as f:
        state_dicts = pickle.load(f)
    return epoch, state_dicts if isinstance(state_dicts, list) else [state_dicts]
