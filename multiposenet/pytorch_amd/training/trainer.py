"""Trainer — the reference's training harness (``training/trainer.py:22-380``) over the MI355X path.

Same construction and behaviour as the reference:

    trainer = Trainer(model, train_params, batch_processor, train_data, val_data)
    trainer.train()

  * ``TrainParams`` carries the reference's fields (:44-103);
  * per epoch: ``_LRScheduler.step(epoch)`` before the epoch, ``_train_one_epoch`` (:233-283: batch_processor -> forward ->
    build_loss -> zero_grad -> backward -> optional gradient clipping -> step, meters, the ``fps`` log line every
    ``print_freq`` steps, step checkpoints), checkpoint ``ckpt_{epoch}.h5`` every ``save_freq_epoch`` epochs (+ pruning),
    validation over ``val_nbatch_end_epoch`` batches in eval mode with ``freeze_bn`` re-applied afterwards unless the
    subnet is the keypoint one (:285-320), a copy ``ckpt_{epoch}_{loss:.5f}.h5.best`` when the validation loss improves and
    ``ReduceLROnPlateau.step(val_loss)`` (:176-217);
  * resume: the newest ``ckpt_*.h5`` of ``save_dir`` (or ``params.ckpt``) restores weights, epoch and optimizer state (:160-175,
    :224-231).
Differences, all forced by the platform: one process per GPU instead of ``ListDataParallel`` (the model goes to
``cuda:gpus[0]``; when ``torch.distributed`` is initialised the gradient reducer of ddp.py is attached and
``batch_size`` is the per-rank batch), log values may be ``numbers.Real`` proxies (losses.set_lazy_log), and
``params.use_graph`` replays the step as a captured hipGraph (graph.py) — off by default.
"""
import datetime
import logging
import numbers
import os
import shutil
import sys
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
from torch.optim.lr_scheduler import ReduceLROnPlateau
from torch.optim.optimizer import Optimizer

try:                                        # torch >= 2.0 names the base class LRScheduler
    from torch.optim.lr_scheduler import LRScheduler as _LRScheduler
except ImportError:                          # pragma: no cover
    from torch.optim.lr_scheduler import _LRScheduler

from ..lib.utils.meter import AverageValueMeter
from ..lib.utils.timer import Timer
from ..network import net_utils

logger = logging.getLogger("multiposenet")


def get_learning_rates(optimizer):
    return np.asarray([pg['lr'] for pg in optimizer.param_groups], dtype=float)


class TrainParams(object):
    # required params (trainer.py:45-50)
    exp_name = 'experiment_name'
    subnet_name = 'keypoint_subnet'
    batch_size = 32
    max_epoch = 30
    optimizer = None
    # learning rate scheduler
    lr_scheduler = None         # ReduceLROnPlateau or an LRScheduler
    max_grad_norm = np.inf
    # local environment
    gpus = [0]
    save_dir = None             # default: outputs/{exp_name}
    # loading an existing checkpoint
    ckpt = None                 # path; None = the newest ckpt in save_dir
    re_init = False
    zero_epoch = False
    ignore_opt_state = False
    # saving checkpoints
    save_freq_epoch = 1
    save_freq_step = sys.maxsize
    save_nckpt_max = sys.maxsize
    # validation during training
    val_freq = 500
    val_nbatch = 10
    val_nbatch_end_epoch = 200
    # logging
    print_freq = 20
    use_tensorboard = False
    visualization_fn = None
    # MI355X path only
    use_graph = False           # replay the train step as a captured hipGraph (graph.GraphedTrainStep)

    def update(self, params_dict):
        for k, v in params_dict.items():
            if hasattr(self, k):
                setattr(self, k, v)
            else:
                logger.warning('Unknown option: {}: {}'.format(k, v))

    def state_dict(self):
        out = OrderedDict()
        for k in TrainParams.__dict__.keys():
            if not k.startswith('_') and k not in ('update', 'state_dict'):
                out[k] = getattr(self, k)
        return out

    def __str__(self):
        return 'TrainParams {\n' + ''.join('\t{}: {}\n'.format(k, v) for k, v in self.state_dict().items()) + '}\n'


class Trainer(object):
    TrainParams = TrainParams
    on_start_epoch_hooks = []
    on_end_epoch_hooks = []

    def __init__(self, model, train_params, batch_processor, train_data, val_data=None):
        assert isinstance(train_params, TrainParams)
        self.params = train_params
        self.train_data = train_data
        self.val_data = val_data
        self.batch_processor = batch_processor
        self.batch_per_epoch = len(self.train_data)
        self.last_epoch = 0
        self.optimizer = self.params.optimizer
        if not isinstance(self.optimizer, Optimizer):
            raise ValueError('optimizer should be an instance of Optimizer, but got {}'.format(type(self.optimizer)))
        self.lr_scheduler = self.params.lr_scheduler
        if self.lr_scheduler and not isinstance(self.lr_scheduler, (ReduceLROnPlateau, _LRScheduler)):
            raise ValueError('lr_scheduler should be an instance of LRScheduler or ReduceLROnPlateau, but got {}'.format(type(self.lr_scheduler)))
        self.log_values = OrderedDict()
        self.batch_timer = Timer()
        self.data_timer = Timer()
        self.model = model
        if not self.params.save_dir:
            self.params.save_dir = os.path.join('outputs', self.params.exp_name)
        os.makedirs(self.params.save_dir, exist_ok=True)
        ckpt = self.params.ckpt
        if ckpt is None:                                       # newest ckpt_*.h5 of save_dir (trainer.py:160-166)
            ckpts = [f for f in os.listdir(self.params.save_dir) if os.path.splitext(f)[-1] == '.h5']
            ckpt = os.path.join(self.params.save_dir, sorted(ckpts, key=lambda n: int(os.path.splitext(n)[0].split('_')[-1]))[-1]) if ckpts else None
        if ckpt is not None and not self.params.re_init:
            self._load_ckpt(ckpt)
            logger.info('Load ckpt from {}'.format(ckpt))
        dev = torch.device('cuda', self.params.gpus[0])
        torch.cuda.set_device(dev)
        self.model = self.model.to(dev)
        if torch.distributed.is_available() and torch.distributed.is_initialized() and getattr(self.model, '_reducer', None) is None:
            from .. import ddp
            ddp.attach(self.model)
        self.model.train()
        if self.params.subnet_name != 'keypoint_subnet':
            self.model.freeze_bn()                             # trainer.py:173-174
        self._graphed = None
        if self.params.use_graph:
            from ..graph import GraphedTrainStep
            self._graphed = GraphedTrainStep(self.model, self.optimizer)

    # ------------------------------------------------------------------ epochs
    def train(self):
        best_loss = np.inf
        for epoch in range(self.last_epoch, self.params.max_epoch):
            self.last_epoch += 1
            logger.info('Start training epoch {}'.format(self.last_epoch))
            for fun in self.on_start_epoch_hooks:
                fun(self)
            if isinstance(self.lr_scheduler, _LRScheduler) and not isinstance(self.lr_scheduler, ReduceLROnPlateau):
                cur_lrs = get_learning_rates(self.optimizer)
                self.lr_scheduler.step(self.last_epoch)
                logger.info('Set learning rates from {} to {}'.format(cur_lrs, get_learning_rates(self.optimizer)))
            self._train_one_epoch()
            for fun in self.on_end_epoch_hooks:
                fun(self)
            if (self.last_epoch % self.params.save_freq_epoch == 0) or (self.last_epoch == self.params.max_epoch - 1):
                save_to = os.path.join(self.params.save_dir, 'ckpt_{}.h5'.format(self.last_epoch))
                self._save_ckpt(save_to)
                if self.params.val_nbatch_end_epoch > 0 and self.val_data is not None:
                    val_loss = self._val_one_epoch(self.params.val_nbatch_end_epoch)
                    if val_loss < best_loss:
                        best_file = os.path.join(self.params.save_dir, 'ckpt_{}_{:.5f}.h5.best'.format(self.last_epoch, val_loss))
                        shutil.copyfile(save_to, best_file)
                        logger.info('Found a better ckpt ({:.5f} -> {:.5f}), saved to {}'.format(best_loss, val_loss, best_file))
                        best_loss = val_loss
                    if isinstance(self.lr_scheduler, ReduceLROnPlateau):
                        self.lr_scheduler.step(val_loss)

    def _save_ckpt(self, save_to):
        model = self.model.module if isinstance(self.model, nn.DataParallel) else self.model
        net_utils.save_net(save_to, model, epoch=self.last_epoch, optimizers=[self.optimizer], rm_prev_opt=True,
                           max_n_ckpts=self.params.save_nckpt_max)
        logger.info('Save ckpt to {}'.format(save_to))

    def _load_ckpt(self, ckpt):
        epoch, state_dicts = net_utils.load_net(ckpt, self.model, load_state_dict=True)
        if not self.params.ignore_opt_state and not self.params.zero_epoch and epoch >= 0:
            self.last_epoch = epoch
            logger.info('Set last epoch to {}'.format(self.last_epoch))
            if state_dicts is not None:
                self.optimizer.load_state_dict(state_dicts[0])
                logger.info('Load optimizer state from checkpoint, new learning rate: {}'.format(get_learning_rates(self.optimizer)))

    def _train_one_epoch(self):
        self.batch_timer.clear()
        self.data_timer.clear()
        self.batch_timer.tic()
        self.data_timer.tic()
        total_loss = AverageValueMeter()
        clip = not np.isinf(self.params.max_grad_norm)
        for step, batch in enumerate(self.train_data):
            inputs, gts, _ = self.batch_processor(self, batch)
            self.data_timer.toc()
            if self._graphed is not None and not clip:
                loss, saved_for_log = self._graphed(inputs, gts)
            else:
                output, saved_for_loss = self.model(*inputs)
                loss, saved_for_log = self.model.build_loss(saved_for_loss, *gts)
                self.optimizer.zero_grad()
                loss.backward()
                if clip:                                       # trainer.py:254-256
                    saved_for_log['max_grad'] = float(nn.utils.clip_grad_norm_(self.model.parameters(), self.params.max_grad_norm, float('inf')))
                self.optimizer.step(None)
            total_loss.add(loss.item())
            self._process_log(saved_for_log, self.log_values)
            self.batch_timer.toc()
            reset = False
            if step % self.params.print_freq == 0:
                self._print_log(step, self.log_values, title='Training', max_n_batch=self.batch_per_epoch)
                reset = True
            if step % self.params.save_freq_step == 0 and step > 0:
                self._save_ckpt(os.path.join(self.params.save_dir, 'ckpt_{}.h5.ckpt'.format((self.last_epoch - 1) * self.batch_per_epoch + step)))
            if reset:
                self._reset_log(self.log_values)
            self.data_timer.tic()
            self.batch_timer.tic()
        return total_loss.value()[0]

    def _val_one_epoch(self, n_batch):
        training_mode = self.model.training
        self.model.eval()
        logs = OrderedDict()
        sum_loss = AverageValueMeter()
        logger.info('Val on validation set...')
        self.batch_timer.clear()
        self.data_timer.clear()
        self.batch_timer.tic()
        self.data_timer.tic()
        with torch.no_grad():
            for step, batch in enumerate(self.val_data):
                self.data_timer.toc()
                if step > n_batch:
                    break
                inputs, gts, _ = self.batch_processor(self, batch)
                _, saved_for_loss = self.model(*inputs)
                self.batch_timer.toc()
                loss, saved_for_log = self.model.build_loss(saved_for_loss, *gts)
                sum_loss.add(loss.item())
                self._process_log(saved_for_log, logs)
                if step % self.params.print_freq == 0 or step == len(self.val_data) - 1:
                    self._print_log(step, logs, 'Validation', max_n_batch=min(n_batch, len(self.val_data)))
                self.data_timer.tic()
                self.batch_timer.tic()
        mean, std = sum_loss.value()
        logger.info('Validation loss: mean: {}, std: {}'.format(mean, std))
        self.model.train(mode=training_mode)
        if self.params.subnet_name != 'keypoint_subnet':
            self.model.freeze_bn()                             # trainer.py:317-318
        return mean

    # ------------------------------------------------------------------ logging
    def _process_log(self, src_dict, dest_dict):
        for k, v in src_dict.items():
            if isinstance(v, numbers.Real) and not isinstance(v, bool):       # floats, ints and LazyFloat proxies (trainer.py:324 tests (int, float))
                dest_dict.setdefault(k, AverageValueMeter())
                dest_dict[k].add(float(v))
            else:
                dest_dict[k] = v

    def _print_log(self, step, log_values, title='', max_n_batch=None):
        log_str = '{}\n'.format(self.params.exp_name)
        log_str += '{}: epoch {}'.format(title, self.last_epoch)
        if max_n_batch:
            log_str += '[{}/{}], lr: {}'.format(step, max_n_batch, get_learning_rates(self.optimizer))
        for k, v in log_values.items():
            if isinstance(v, AverageValueMeter):
                log_str += '\n\t{}: {:.10f}'.format(k, v.value()[0])
        if max_n_batch:
            data_time = self.data_timer.duration + 1e-6
            batch_time = self.batch_timer.duration + 1e-6
            rest_seconds = int((max_n_batch - step) * batch_time)
            log_str += '\n\t({:.2f}/{:.2f}s, fps:{:.1f}, rest: {})'.format(data_time, batch_time, self.params.batch_size / batch_time,
                                                                           str(datetime.timedelta(seconds=rest_seconds)))
            self.batch_timer.clear()
            self.data_timer.clear()
        logger.info(log_str)

    def _reset_log(self, log_values):
        for v in log_values.values():
            if isinstance(v, AverageValueMeter):
                v.reset()
