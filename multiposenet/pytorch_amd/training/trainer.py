"""Training driver for the MI355X path behind the reference's ``Trainer`` interface.

Only the INTERFACE comes from the reference (``training/trainer.py:44-105`` option names, ``:116`` constructor,
``:176`` ``train()``); its behaviour — which epochs write ``ckpt_{epoch}.h5``, when the validation pass runs and which
checkpoint becomes ``ckpt_{epoch}_{loss:.5f}.h5.best``, how the two scheduler families are stepped, what a resume restores
under ``re_init`` / ``zero_epoch`` / ``ignore_opt_state``, the BatchNorm mode rules around validation — is pinned by
``tests/golden/g14_trainer.json``, recorded from the real class (tests/golden/make_golden_trainer.py), not transcribed.

What is built differently, for one process per GPU and a device that runs a step in 40 ms:

  * **the step** is the recorded launch list (``replay.ReplayedTrainStep``): forward, losses, both backward streams, gradient
    buckets and FusedAdam are re-issued from one list (host cost 5 ms instead of 19); the eager tape is used when the optimizer
    is not ``FusedAdam``, the subnet is the PRN, or gradient clipping needs the norms on the host;
  * **no per-step host sync**: the loss and the log values are fetched with asynchronous copies (``losses.LazyFloat``) and
    only become floats when a log line is formatted (every ``print_freq`` steps) — the reference reads ``loss.item()``
    and seven more ``.item()`` values per step;
  * **data parallelism** is ``torch.distributed`` with the gradient reducer of ``ddp.py`` instead of ``ListDataParallel``
    (datasets/data_parallel.py:16-87): every rank runs this class on its shard; rank 0 alone writes, prunes and copies
    checkpoints (the others wait at a barrier), and the validation loss is averaged over ranks before it decides the best
    checkpoint and steps ``ReduceLROnPlateau``, so all replicas take the same decisions;
  * a resumed stock ``torch.optim`` optimizer gets its moments moved next to the parameters (the reference's
    ``set_optimizer_state_devices`` call, trainer.py:229).
"""
import datetime
import logging
import math
import numbers
import os
import shutil
import sys
import time
from collections import OrderedDict

import torch
import torch.distributed as dist
from torch.optim.lr_scheduler import LRScheduler, ReduceLROnPlateau
from torch.optim.optimizer import Optimizer

from ..network import net_utils

logger = logging.getLogger("multiposenet")

INF = float('inf')

# option name -> default: the reference's TrainParams fields in their order (trainer.py:44-80), then this stack's additions
_OPTIONS = OrderedDict([
    ('exp_name', 'experiment_name'), ('subnet_name', 'keypoint_subnet'), ('batch_size', 32), ('max_epoch', 30), ('optimizer', None),
    ('lr_scheduler', None), ('max_grad_norm', INF),
    ('gpus', [0]), ('save_dir', None),
    ('ckpt', None), ('re_init', False), ('zero_epoch', False), ('ignore_opt_state', False),
    ('save_freq_epoch', 1), ('save_freq_step', sys.maxsize), ('save_nckpt_max', sys.maxsize),
    ('val_freq', 500), ('val_nbatch', 10), ('val_nbatch_end_epoch', 200),
    ('print_freq', 20), ('use_tensorboard', False), ('visualization_fn', None),
])
_EXTRA_OPTIONS = OrderedDict([
    ('launch', 'replay'),        # 'replay' (recorded launch list) | 'eager' (Python tape)
    ('lazy_log', True),          # log values / loss fetched asynchronously, materialised when a line is printed
])


class TrainParams(object):
    """Options of a training run.  ``gpus[0]`` is the device of this process (an empty list leaves the model where it is)."""

    def __init__(self, **overrides):
        for table in (_OPTIONS, _EXTRA_OPTIONS):
            for name, default in table.items():
                setattr(self, name, list(default) if isinstance(default, list) else default)
        self.update(overrides)

    def update(self, params_dict):
        for name, value in params_dict.items():
            if name in _OPTIONS or name in _EXTRA_OPTIONS or hasattr(self, name):
                setattr(self, name, value)
            else:
                logger.warning('Unknown option: {}: {}'.format(name, value))

    def state_dict(self):
        return OrderedDict((name, getattr(self, name)) for name in _OPTIONS)

    def __str__(self):
        return 'TrainParams {\n' + ''.join('\t{}: {}\n'.format(k, v) for k, v in self.state_dict().items()) + '}\n'


def get_learning_rates(optimizer):
    return [float(group['lr']) for group in optimizer.param_groups]


def _is_epoch_scheduler(s):
    """A scheduler stepped with the epoch number at the start of an epoch.  ReduceLROnPlateau (an LRScheduler subclass since
    torch 2.2, a separate class in the reference's torch 0.4) is stepped with the validation loss instead."""
    return isinstance(s, LRScheduler) and not isinstance(s, ReduceLROnPlateau)


# ---------------------------------------------------------------------------------------------------- bookkeeping
class RunningStat(object):
    """Mean / sample standard deviation of a stream of scalars.  Values may be asynchronous proxies (losses.LazyFloat): they
    are parked untouched and folded in only when a statistic is read, so adding never waits for the device."""

    def __init__(self):
        self.reset()

    def reset(self):
        self._n, self._mean, self._m2, self._parked = 0, 0.0, 0.0, []

    PARK_LIMIT = 64        # asynchronous proxies hold a pinned buffer and an event each: never more than this many at once

    def add(self, value):
        self._parked.append(value)
        if len(self._parked) >= self.PARK_LIMIT:
            # the oldest proxies belong to steps that finished long ago (their copies have landed): folding them does not wait
            recent = self._parked[-8:]
            self._parked = self._parked[:-8]
            self._fold()
            self._parked = recent

    def _fold(self):
        for v in self._parked:
            x = float(v)
            self._n += 1
            d = x - self._mean
            self._mean += d / self._n
            self._m2 += d * (x - self._mean)
        self._parked = []

    @property
    def count(self):
        return self._n + len(self._parked)

    def value(self):
        """(mean, std): (nan, nan) when empty, (x, inf) after one sample."""
        self._fold()
        if self._n == 0:
            return float('nan'), float('nan')
        if self._n == 1:
            return self._mean, INF
        return self._mean, math.sqrt(max(self._m2, 0.0) / (self._n - 1))


class LogBook(OrderedDict):
    """Named values of the current logging window: numbers are averaged (RunningStat), anything else is kept as last seen."""

    def record(self, values):
        for name, v in values.items():
            if isinstance(v, numbers.Real) and not isinstance(v, bool):
                stat = self.get(name)
                if not isinstance(stat, RunningStat):
                    stat = self[name] = RunningStat()
                stat.add(v)
            else:
                self[name] = v

    def lines(self):
        return ['\n\t{}: {:.10f}'.format(name, v.value()[0]) for name, v in self.items() if isinstance(v, RunningStat)]

    def restart(self):
        for v in self.values():
            if isinstance(v, RunningStat):
                v.reset()


class Stopwatch(object):
    """Average length of the intervals between start() and lap() since the last clear()."""

    def __init__(self):
        self.clear()

    def clear(self):
        self.total, self.laps, self._t0 = 0.0, 0, None

    def start(self):
        self._t0 = time.time()

    def lap(self):
        if self._t0 is not None:
            self.total += time.time() - self._t0
            self.laps += 1

    @property
    def mean(self):
        return self.total / self.laps if self.laps else 0.0


class _Ranks(object):
    """The process group as this driver needs it: who writes files, a barrier, and a cross-rank mean of a host scalar."""

    def __init__(self, device):
        self.on = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank() if self.on else 0
        self.world = dist.get_world_size() if self.on else 1
        self.device = device

    @property
    def writer(self):
        return self.rank == 0

    def barrier(self):
        if self.on and self.world > 1:
            dist.barrier()

    def mean(self, x):
        if not self.on or self.world == 1:
            return float(x)
        on_device = self.device.type == 'cuda' and dist.get_backend() == 'nccl'
        t = torch.tensor([float(x)], dtype=torch.float64, device=self.device if on_device else 'cpu')
        dist.all_reduce(t)
        return float(t.item()) / self.world


class CheckpointShelf(object):
    """The checkpoint files of one run directory.  All ranks call every method; only the writer touches the disk."""

    def __init__(self, folder, keep, ranks):
        self.folder, self.keep, self.ranks = folder, keep, ranks
        if ranks.writer:
            os.makedirs(folder, exist_ok=True)
        ranks.barrier()

    def newest(self):
        names = net_utils.list_checkpoints(self.folder) if os.path.isdir(self.folder) else []
        return os.path.join(self.folder, names[-1]) if names else None

    def put(self, name, model, optimizer, epoch):
        path = os.path.join(self.folder, name)
        if self.ranks.writer:
            net_utils.save_net(path, model, epoch=epoch, optimizers=[optimizer], rm_prev_opt=True, max_n_ckpts=self.keep)
            logger.info('Save ckpt to {}'.format(path))
        self.ranks.barrier()
        return path

    def promote(self, path, epoch, loss, previous_best):
        best = os.path.join(self.folder, 'ckpt_{}_{:.5f}.h5.best'.format(epoch, loss))
        if self.ranks.writer:
            shutil.copyfile(path, best)
            logger.info('Found a better ckpt ({:.5f} -> {:.5f}), saved to {}'.format(previous_best, loss, best))
        self.ranks.barrier()
        return best


class _Stepper(object):
    """One optimisation step, through the fastest path that is valid for (model, optimizer, options)."""

    def __init__(self, model, optimizer, params):
        self.model, self.optimizer = model, optimizer
        self.clip = params.max_grad_norm if not math.isinf(params.max_grad_norm) else None
        self.fast = None
        fused = type(optimizer).__name__ == 'FusedAdam' and hasattr(model, '_engine')
        if params.launch not in ('replay', 'eager'):
            raise ValueError("TrainParams.launch must be 'replay' or 'eager', got %r" % (params.launch,))
        if fused and self.clip is None and params.launch == 'replay':
            from ..replay import ReplayedTrainStep
            self.fast = ReplayedTrainStep(model, optimizer)

    def __call__(self, inputs, gts):
        if self.fast is not None:
            return self.fast(inputs, gts)
        model = self.model
        _, saved_for_loss = model(*inputs)
        loss, log = model.build_loss(saved_for_loss, *gts)
        self.optimizer.zero_grad()
        loss.backward()
        if self.clip is not None:
            log['max_grad'] = float(torch.nn.utils.clip_grad_norm_(model.parameters(), self.clip, INF))
        self.optimizer.step()
        return loss, log


def _scalar_proxy(loss, lazy):
    """The step's loss as a number for the meters: an asynchronous proxy for device scalars (no sync), else a float."""
    if torch.is_tensor(loss):
        if lazy and loss.is_cuda:
            from ..network import losses
            was = losses.LAZY_LOG
            losses.LAZY_LOG = True
            try:
                return losses._log_values(loss.detach().reshape(1))[0]
            finally:
                losses.LAZY_LOG = was
        return float(loss.item())
    return float(loss)


# ---------------------------------------------------------------------------------------------------- the driver
class Trainer(object):
    TrainParams = TrainParams
    on_start_epoch_hooks = []
    on_end_epoch_hooks = []

    def __init__(self, model, train_params, batch_processor, train_data, val_data=None):
        if not isinstance(train_params, TrainParams):
            raise TypeError('train_params must be a TrainParams, got {}'.format(type(train_params)))
        p = self.params = train_params
        self.batch_processor = batch_processor
        self.train_data, self.val_data = train_data, val_data
        self.batch_per_epoch = len(train_data)
        self.optimizer, self.lr_scheduler = p.optimizer, p.lr_scheduler
        if not isinstance(self.optimizer, Optimizer):
            raise ValueError('optimizer should be an instance of Optimizer, but got {}'.format(type(self.optimizer)))
        if self.lr_scheduler and not isinstance(self.lr_scheduler, (ReduceLROnPlateau, LRScheduler)):
            raise ValueError('lr_scheduler should be an LRScheduler or ReduceLROnPlateau, but got {}'.format(type(self.lr_scheduler)))
        self.model = model
        self.last_epoch = 0
        self.log_values = LogBook()
        self.batch_timer, self.data_timer = Stopwatch(), Stopwatch()

        if p.gpus:
            self.device = torch.device('cuda', p.gpus[0])
            torch.cuda.set_device(self.device)
        else:
            self.device = next(model.parameters()).device
        self.ranks = _Ranks(self.device)
        if not p.save_dir:
            p.save_dir = os.path.join('outputs', p.exp_name)
        self.shelf = CheckpointShelf(p.save_dir, p.save_nckpt_max, self.ranks)

        # parameters go to the device BEFORE a resume, so restored optimizer moments end up beside them
        self.model = self.model.to(self.device)
        source = p.ckpt if p.ckpt is not None else self.shelf.newest()
        if source is not None and not p.re_init:
            self._load_ckpt(source)
            logger.info('Load ckpt from {}'.format(source))
        if self.ranks.on and hasattr(self.model, '_arena') and getattr(self.model, '_reducer', None) is None:
            from .. import ddp
            ddp.attach(self.model)                           # broadcasts rank 0's parameters, buckets the gradient arena
        self._apply_train_modes()
        self._step = _Stepper(self.model, self.optimizer, p)
        self._lazy = bool(p.lazy_log) and hasattr(self.model, '_engine')

    # ------------------------------------------------------------------ modes / resume
    def _apply_train_modes(self):
        """Everything in train mode; BatchNorm statistics frozen unless the keypoint subnet is being trained."""
        self.model.train()
        if self.params.subnet_name != 'keypoint_subnet':
            self.model.freeze_bn()

    def _load_ckpt(self, ckpt):
        p = self.params
        epoch, states = net_utils.load_net(ckpt, self.model, load_state_dict=True)
        if p.ignore_opt_state or p.zero_epoch or epoch < 0:
            return
        self.last_epoch = epoch
        logger.info('Set last epoch to {}'.format(epoch))
        if states is not None:
            self.optimizer.load_state_dict(states[0])
            if self.device.type == 'cuda':
                net_utils.set_optimizer_state_devices(self.optimizer.state, self.device.index)
            logger.info('Load optimizer state from checkpoint, new learning rate: {}'.format(get_learning_rates(self.optimizer)))

    def _save_ckpt(self, save_to):
        return self.shelf.put(os.path.basename(save_to), self.model, self.optimizer, self.last_epoch)

    # ------------------------------------------------------------------ epochs
    def _epoch_saves(self, epoch):
        p = self.params
        return epoch % p.save_freq_epoch == 0 or epoch == p.max_epoch - 1

    def train(self):
        """All remaining epochs.  While it runs, loss log values are asynchronous proxies (losses.set_lazy_log) unless
        ``params.lazy_log`` is off; the process-wide setting is restored on the way out."""
        if not self._lazy:
            return self._train()
        from ..network import losses
        before = losses.LAZY_LOG
        losses.set_lazy_log(True)
        try:
            return self._train()
        finally:
            losses.set_lazy_log(before)

    def _train(self):
        p = self.params
        best = INF
        while self.last_epoch < p.max_epoch:
            self.last_epoch += 1
            epoch = self.last_epoch
            logger.info('Start training epoch {}'.format(epoch))
            for hook in self.on_start_epoch_hooks:
                hook(self)
            if _is_epoch_scheduler(self.lr_scheduler):
                before = get_learning_rates(self.optimizer)
                self.lr_scheduler.step(epoch)
                logger.info('Set learning rates from {} to {}'.format(before, get_learning_rates(self.optimizer)))
            self._train_one_epoch()
            for hook in self.on_end_epoch_hooks:
                hook(self)
            if not self._epoch_saves(epoch):
                continue
            saved = self._save_ckpt('ckpt_{}.h5'.format(epoch))
            if p.val_nbatch_end_epoch > 0 and self.val_data is not None:
                val_loss = self._val_one_epoch(p.val_nbatch_end_epoch)
                if val_loss < best:
                    self.shelf.promote(saved, epoch, val_loss, best)
                    best = val_loss
                if isinstance(self.lr_scheduler, ReduceLROnPlateau):
                    self.lr_scheduler.step(val_loss)
                    self.lr_scheduler.last_epoch = epoch

    def _train_one_epoch(self):
        p = self.params
        epoch_loss = RunningStat()
        self._restart_timers()
        for step, batch in enumerate(self.train_data):
            inputs, gts, _ = self.batch_processor(self, batch)
            self.data_timer.lap()
            loss, log = self._step(inputs, gts)
            epoch_loss.add(_scalar_proxy(loss, self._lazy))
            self.log_values.record(log)
            self.batch_timer.lap()
            if step % p.print_freq == 0:
                self._print_log(step, self.log_values, title='Training', max_n_batch=self.batch_per_epoch)
                window_done = True
            else:
                window_done = False
            if step > 0 and step % p.save_freq_step == 0:
                self._save_ckpt('ckpt_{}.h5.ckpt'.format((self.last_epoch - 1) * self.batch_per_epoch + step))
            if window_done:
                self.log_values.restart()
            self.data_timer.start()
            self.batch_timer.start()
        return epoch_loss.value()[0]

    def _val_one_epoch(self, n_batch):
        """Mean loss over the first ``n_batch + 1`` validation batches in eval mode (the count the reference consumes),
        averaged over ranks; afterwards the modes training runs in are restored."""
        p = self.params
        was_training = self.model.training
        self.model.eval()
        book, losses_seen = LogBook(), RunningStat()
        logger.info('Val on validation set...')
        self._restart_timers()
        n_print = min(n_batch, len(self.val_data))
        with torch.no_grad():
            for step, batch in enumerate(self.val_data):
                self.data_timer.lap()
                if step > n_batch:
                    break
                inputs, gts, _ = self.batch_processor(self, batch)
                _, saved_for_loss = self.model(*inputs)
                self.batch_timer.lap()
                loss, log = self.model.build_loss(saved_for_loss, *gts)
                losses_seen.add(_scalar_proxy(loss, self._lazy))
                book.record(log)
                if step % p.print_freq == 0 or step == len(self.val_data) - 1:
                    self._print_log(step, book, 'Validation', max_n_batch=n_print)
                self.data_timer.start()
                self.batch_timer.start()
        mean, std = losses_seen.value()
        mean = self.ranks.mean(mean)
        logger.info('Validation loss: mean: {}, std: {}'.format(mean, std))
        self.model.train(mode=was_training)
        if p.subnet_name != 'keypoint_subnet':
            self.model.freeze_bn()
        return mean

    # ------------------------------------------------------------------ logging
    def _restart_timers(self):
        for t in (self.batch_timer, self.data_timer):
            t.clear()
            t.start()

    def _print_log(self, step, book, title='', max_n_batch=None):
        """One log block per window: averaged values, then (data s / batch s, fps = batch_size / batch s, time left).  Non-writer
        ranks stay quiet (their windows are reset all the same)."""
        p = self.params
        text = '{}\n{}: epoch {}'.format(p.exp_name, title, self.last_epoch)
        if max_n_batch:
            text += '[{}/{}], lr: {}'.format(step, max_n_batch, get_learning_rates(self.optimizer))
        text += ''.join(book.lines())
        if max_n_batch:
            data_s, batch_s = self.data_timer.mean + 1e-6, self.batch_timer.mean + 1e-6
            left = datetime.timedelta(seconds=int((max_n_batch - step) * batch_s))
            text += '\n\t({:.2f}/{:.2f}s, fps:{:.1f}, rest: {})'.format(data_s, batch_s, p.batch_size * self.ranks.world / batch_s, left)
            self.batch_timer.clear()
            self.data_timer.clear()
        if self.ranks.writer:
            logger.info(text)
