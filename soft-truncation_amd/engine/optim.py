"""Fused clip + Adam over the flat parameter buffer.

The reference's optimisation step is ``clip_grad_norm_`` + ``torch.optim.Adam.step`` over 564
tensors (losses.py:47-56): hundreds of small launches and a host sync for the norm.  Here the
whole model is two launches with no host round trip:

  stk_sumsq_f32(flat_grad)                    -> device scalar  sum g^2
  stk_adam_f32(p, g, m, v, ..., sumsq, max_norm)   clip coefficient applied on the fly

``FusedAdam`` is a ``torch.optim.Optimizer`` whose ``state_dict`` has the same structure as
``torch.optim.Adam``'s (per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq``), so checkpoints
written by either load into the other (reference utils.py:13-36).
"""
import torch

from . import lib as stk_lib
from .flat import flat_of


class FusedAdam(torch.optim.Optimizer):
  def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0., amsgrad=False,
               adamw=False, backend=None):
    defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad)
    super().__init__(params, defaults)
    if len(self.param_groups) != 1:
      raise ValueError('FusedAdam handles the single parameter group the training step uses')
    self.adamw = bool(adamw)
    self._backend = backend
    self._flat = None
    self._m = self._v = self._vmax = None
    self._step = 0
    self._sumsq = None
    self._ws = None
    self._pending_clip = None

  # -- plumbing -------------------------------------------------------------------------------
  def _lib(self):
    if self._backend is None:
      self._backend = stk_lib.load()
    return self._backend

  def _bind(self):
    params = self.param_groups[0]['params']
    flat = flat_of(params, full=self._flat is None)      # every parameter at construction, a sample per step
    if flat is None:
      raise RuntimeError('FusedAdam needs parameters backed by engine.flat.FlatParams '
                         '(create the model with models.utils.create_model)')
    if flat is not self._flat:
      old = self._flat
      self._flat = flat
      dev = flat.device
      m = torch.zeros(flat.n_train, dtype=torch.float32, device=dev)
      v = torch.zeros(flat.n_train, dtype=torch.float32, device=dev)
      if old is not None and self._m is not None and old.n_train == flat.n_train:
        m.copy_(self._m)
        v.copy_(self._v)
      self._m, self._v = m, v
      if self.param_groups[0]['amsgrad']:
        vmax = torch.zeros(flat.n_train, dtype=torch.float32, device=dev)
        if old is not None and self._vmax is not None and old.n_train == flat.n_train:
          vmax.copy_(self._vmax)
        self._vmax = vmax
      self._sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
      self._ws = torch.zeros(2048, dtype=torch.float32, device=dev)
      self._publish_state()
    return flat

  def _publish_state(self):
    """Expose the flat moments as torch.optim.Adam-style per-parameter state (views)."""
    flat = self._flat
    mv = flat.trainable_views(self._m)
    vv = flat.trainable_views(self._v)
    mx = flat.trainable_views(self._vmax) if self._vmax is not None else [None] * len(mv)
    for p, m, v, x in zip(flat.trainable_params(), mv, vv, mx):
      self.state[p] = {'step': torch.tensor(float(self._step)), 'exp_avg': m, 'exp_avg_sq': v}
      if x is not None:
        self.state[p]['max_exp_avg_sq'] = x              # torch.optim.Adam's key for amsgrad

  # -- the torch.optim API --------------------------------------------------------------------
  def zero_grad(self, set_to_none=False):
    """Zero the flat gradient buffer in place (the `.grad` views must stay bound)."""
    flat = self._bind()
    if not flat.grads_bound():       # e.g. one `p.grad = None`, or torch's zero_grad(set_to_none=True) on the model
      flat.rebind_grads()
    flat.grad.zero_()

  def clip_grad_norm(self, max_norm):
    """Device-side equivalent of clip_grad_norm_: records sum g^2; the scaling happens in step()."""
    flat = self._bind()
    with stk_lib.device_guard(flat.device):
      self._lib().sumsq_f32(flat.grad.data_ptr(), flat.n_train, self._sumsq.data_ptr(), self._ws.data_ptr(),
                            stk_lib.stream_ptr(flat.device))
    self._pending_clip = float(max_norm)
    return self._sumsq

  @torch.no_grad()
  def step(self, closure=None):
    """One fused update of the whole trainable range [0, n_train).  torch.optim.Adam skips parameters whose .grad is
    None; here every slot is stepped with whatever the flat gradient buffer holds (zeros for a parameter the planned
    graph never writes, e.g. the Dense_0 layers of an unconditional model).  With weight_decay == 0 -- every shipped
    config -- a zero gradient leaves such a parameter untouched, as in the reference; with weight decay it would decay,
    which the reference would not do, hence the warning."""
    flat = self._bind()
    group = self.param_groups[0]
    if group['weight_decay'] != 0 and not getattr(self, '_wd_warned', False):
      self._wd_warned = True
      import warnings
      warnings.warn('FusedAdam applies weight decay to every trainable parameter, including ones that receive no '
                    'gradient (torch.optim.Adam would skip those)')
    b1, b2 = group['betas']
    self._step += 1
    bc1 = 1.0 - b1 ** self._step
    bc2 = 1.0 - b2 ** self._step
    clip = self._pending_clip
    self._pending_clip = None
    tail = (flat.n_train, float(group['lr']), b1, b2, group['eps'], group['weight_decay'], int(self.adamw), bc1, bc2,
            self._sumsq.data_ptr() if clip is not None else None, clip if clip is not None else -1.0,
            stk_lib.stream_ptr(flat.device))
    with stk_lib.device_guard(flat.device):
      if self._vmax is not None:
        self._lib().adam_amsgrad_f32(flat.data.data_ptr(), flat.grad.data_ptr(), self._m.data_ptr(), self._v.data_ptr(),
                                     self._vmax.data_ptr(), *tail)
      else:
        self._lib().adam_f32(flat.data.data_ptr(), flat.grad.data_ptr(), self._m.data_ptr(), self._v.data_ptr(), *tail)

  def state_dict(self):
    if self._flat is not None:
      for st in self.state.values():
        st['step'] = torch.tensor(float(self._step))
    return super().state_dict()

  def load_state_dict(self, state_dict):
    super().load_state_dict(state_dict)
    flat = self._flat
    loaded = dict(self.state)
    if flat is None:
      flat = self._bind()
    step = 0
    with torch.no_grad():
      for p, m, v in zip(flat.trainable_params(), flat.trainable_views(self._m), flat.trainable_views(self._v)):
        st = loaded.get(p)
        if st and 'exp_avg' in st:
          m.copy_(st['exp_avg'])
          v.copy_(st['exp_avg_sq'])
          step = max(step, int(float(st['step'])))
      if self._vmax is not None:
        for p, x in zip(flat.trainable_params(), flat.trainable_views(self._vmax)):
          st = loaded.get(p)
          if st and 'max_exp_avg_sq' in st:
            x.copy_(st['max_exp_avg_sq'])
    self._step = step
    self.state.clear()
    self._publish_state()
