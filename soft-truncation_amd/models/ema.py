"""Exponential moving average of the parameters (reference: models/ema.py:10-98).

Same interface and ``state_dict`` layout (``decay``, ``num_updates``, ``shadow_params`` list).
When the parameters are backed by the engine's flat buffer, the shadow copy is one flat buffer
too and ``update`` is a single ``stk_ema_f32`` launch instead of a Python loop over 564 tensors
(models/ema.py:50-51); ``shadow_params`` are views into it.  Parameter lists that are not
flat-backed (anything that is not the score network) use the same formula tensor by tensor.
"""
import torch

from ..engine import lib as stk_lib
from ..engine.flat import flat_of


class ExponentialMovingAverage:
  def __init__(self, parameters, decay, use_num_updates=True):
    if decay < 0.0 or decay > 1.0:
      raise ValueError('Decay must be between 0 and 1')
    self.decay = decay
    self.num_updates = 0 if use_num_updates else None
    parameters = list(parameters)
    self._flat = flat_of(parameters) if parameters else None
    self._backend = None
    if self._flat is not None:
      self._shadow = self._flat.data[:self._flat.n_train].clone()
      self.shadow_params = self._flat.trainable_views(self._shadow)
    else:
      self._shadow = None
      self.shadow_params = [p.clone().detach() for p in parameters if p.requires_grad]
    self.collected_params = []

  def set_backend(self, backend):
    self._backend = backend

  def _lib(self):
    if self._backend is None:
      self._backend = stk_lib.load()
    return self._backend

  def update(self, parameters):
    """s <- s - (1 - d)(s - p) with the warm-up decay d = min(decay, (1+n)/(10+n)) (:43-51)."""
    decay = self.decay
    if self.num_updates is not None:
      self.num_updates += 1
      decay = min(decay, (1 + self.num_updates) / (10 + self.num_updates))
    one_minus_decay = 1.0 - decay
    with torch.no_grad():
      if self._flat is not None:
        # the flat buffer the parameters live in NOW: the first parameter stands for all (a model is moved as a whole;
        # listing the module tree is ~0.9 ms of host time per step)
        first = next(iter(parameters), None)
        live = getattr(first, '_stk_flat', None) if first is not None else self._flat
        if live is not None and live is not self._flat:
          # the executor rebuilt its flat buffer (a .to() / dtype move): follow it, keeping the averaged values
          old = [s.clone() for s in self.shadow_params]
          self._flat = live
          self._shadow = live.data[:live.n_train].clone()
          self.shadow_params = live.trainable_views(self._shadow)
          for s_new, s_old in zip(self.shadow_params, old):
            s_new.copy_(s_old.to(s_new.device))
        elif live is None:
          raise RuntimeError('ExponentialMovingAverage.update: the parameters no longer live in an engine.flat.FlatParams '
                             'buffer; rebuild the average from the current parameters')
        flat = self._flat
        with stk_lib.device_guard(flat.device):
          self._lib().ema_f32(self._shadow.data_ptr(), flat.data.data_ptr(), flat.n_train, one_minus_decay,
                              stk_lib.stream_ptr(flat.device))
        return
      parameters = [p for p in parameters if p.requires_grad]
      for s_param, param in zip(self.shadow_params, parameters):
        s_param.sub_(one_minus_decay * (s_param - param))

  def copy_to(self, parameters):
    parameters = [p for p in parameters if p.requires_grad]
    for s_param, param in zip(self.shadow_params, parameters):
      if param.requires_grad:
        param.data.copy_(s_param.data)

  def store(self, parameters):
    self.collected_params = [param.clone() for param in parameters]

  def restore(self, parameters):
    for c_param, param in zip(self.collected_params, parameters):
      param.data.copy_(c_param.data)

  def state_dict(self):
    return dict(decay=self.decay, num_updates=self.num_updates, shadow_params=self.shadow_params)

  def load_state_dict(self, state_dict):
    self.decay = state_dict['decay']
    self.num_updates = state_dict['num_updates']
    if self._flat is not None:
      with torch.no_grad():
        for s, loaded in zip(self.shadow_params, state_dict['shadow_params']):
          s.copy_(loaded)
    else:
      self.shadow_params = state_dict['shadow_params']
