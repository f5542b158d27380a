"""Driver glue on the boundary of the hot path (reference: utils.py:13-82).

``restore_checkpoint`` / ``save_checkpoint`` keep the reference's on-disk layout -- a
``torch.save`` of ``{'optimizer', 'model', 'ema', 'step'}`` with ``module.``-prefixed model keys and
``ema.shadow_params`` as a list of tensors -- so checkpoints are interchangeable.
``load_model`` and ``get_loss_fns`` reproduce what ``run_lib.train`` calls before its loop
(run_lib.py:51-66).  The TensorFlow file API of the reference (``tf.io.gfile``) is replaced by
``os``.
"""
import logging
import os

import numpy as np
import torch

from . import likelihood, losses, sampling
from .models import utils as mutils
from .models.ema import ExponentialMovingAverage


_STATEFUL = ('optimizer', 'model', 'ema')          # entries of `state` with a state_dict; 'step' is a plain int


def restore_checkpoint(config, ckpt_dir, state, device):
  """Load {'optimizer', 'model', 'ema', 'step'} into `state` in place; a missing file leaves it as it is
  (utils.py:13-25)."""
  if not os.path.exists(ckpt_dir):
    os.makedirs(os.path.dirname(ckpt_dir), exist_ok=True)
    logging.warning(f"No checkpoint found at {ckpt_dir}. Returned the same state as input")
    return state
  logging.info(ckpt_dir + ' loaded ...')
  loaded = torch.load(ckpt_dir, map_location=device, weights_only=False)
  for key in _STATEFUL:
    kwargs = dict(strict=False) if key == 'model' else {}
    state[key].load_state_dict(loaded[key], **kwargs)
  state['step'] = loaded['step']
  return state


def save_checkpoint(config, ckpt_dir, state):
  """utils.py:28-36."""
  torch.save({**{key: state[key].state_dict() for key in _STATEFUL}, 'step': state['step']}, ckpt_dir)


def load_model(config, workdir, print_=True, sde=None):
  """Model + optimizer + EMA + (resumed) step (utils.py:49-73)."""
  score_model = mutils.create_model(config, sde)
  optimizer = losses.get_optimizer(config, score_model.parameters())
  ema = ExponentialMovingAverage(score_model.parameters(), decay=config.model.ema_rate)
  state = dict(optimizer=optimizer, model=score_model, ema=ema, step=0)
  if print_:
    sizes = [(np.prod(p.size()), p.requires_grad) for p in score_model.parameters()]
    logging.info(f"model parameters: {sum(n for n, trainable in sizes if trainable)}")
    logging.info(f"total number of parameters: {sum(n for n, _ in sizes)}")
  checkpoint_dir = os.path.join(workdir, "checkpoints")
  checkpoint_meta_dir = os.path.join(workdir, "checkpoints-meta", "checkpoint.pth")
  os.makedirs(checkpoint_dir, exist_ok=True)
  os.makedirs(os.path.dirname(checkpoint_meta_dir), exist_ok=True)
  state = restore_checkpoint(config, checkpoint_meta_dir, state, config.device)
  return state, score_model, ema, checkpoint_dir, checkpoint_meta_dir


def get_loss_fns(config, sde, inverse_scaler, train=True):
  """(train_step_fn, nll_fn, nelbo_fn, sampling_fn) as in utils.py:75-82."""
  optimize_fn = losses.optimization_manager(config)
  train_step_fn = losses.get_step_fn(config, sde, train=train, optimize_fn=optimize_fn)
  sampling_shape = (config.sampling.batch_size, config.data.num_channels,
                    config.data.image_size, config.data.image_size)
  sampling_fn = sampling.get_sampling_fn(config, sde, sampling_shape, inverse_scaler,
                                         config.sampling.truncation_time)
  nll_fn = likelihood.get_likelihood_fn(config, sde, inverse_scaler)
  nelbo_fn = likelihood.get_elbo_fn(config, sde, inverse_scaler=inverse_scaler)
  return train_step_fn, nll_fn, nelbo_fn, sampling_fn
