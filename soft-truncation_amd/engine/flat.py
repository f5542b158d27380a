"""Flat parameter / gradient storage.

All trainable parameters of the score network live in ONE contiguous fp32 buffer (and their
gradients in a second one) laid out for the device rather than for the module tree:

* the fused clip + Adam + EMA kernels stream the whole model in three launches instead of the
  reference's Python loops over 564 tensors (losses.py:55-56, models/ema.py:50-51);
* the RCCL gradient all-reduce works on a handful of large contiguous buckets;
* the per-block time-embedding projections (46 x nn.Linear(512, Cout), models/layerspp.py:240)
  are placed back to back so the engine evaluates them as one [B,512]x[512,sum Cout] GEMM.

``nn.Parameter`` objects stay what the reference's code expects (``state_dict`` keys, shapes,
``.grad``) -- their ``.data`` / ``.grad`` are views into the flat buffers.
"""
import torch

ALIGN = 64  # floats


def _round_up(n, a=ALIGN):
  return (n + a - 1) // a * a


class FlatParams:
  def __init__(self, params, device, groups=()):
    """params: ordered list of nn.Parameter.  groups: lists of parameters that must be laid out
    contiguously and unpadded, in order (e.g. all Dense_0 weights)."""
    self.device = torch.device(device)
    params = list(params)
    self.module_order = params      # the order of model.parameters(): what state_dicts / EMA lists use
    seen = set()
    order = []
    self.group_ranges = []
    for grp in groups:
      grp = [p for p in grp]
      for p in grp:
        assert id(p) not in seen
        seen.add(id(p))
      order.append(grp)
    for p in params:
      if id(p) not in seen:
        seen.add(id(p))
        order.append([p])
    # trainable first, frozen after (EMA / Adam / all-reduce cover [0, n_train))
    trainable = [g for g in order if g[0].requires_grad]
    frozen = [g for g in order if not g[0].requires_grad]
    self._slot = {}
    off = 0
    self.params = []
    for grp in trainable:
      start = off
      for p in grp:
        self._slot[id(p)] = (off, p.numel(), True)
        self.params.append(p)
        off += p.numel()
      self.group_ranges.append((start, off))
      off = _round_up(off)
    self.n_train = off
    for grp in frozen:
      for p in grp:
        self._slot[id(p)] = (off, p.numel(), False)
        self.params.append(p)
        off += p.numel()
      off = _round_up(off)
    self.n_total = off
    self.data = torch.zeros(self.n_total, dtype=torch.float32, device=self.device)
    self.grad = torch.zeros(max(self.n_train, 1), dtype=torch.float32, device=self.device)
    with torch.no_grad():
      for p in self.params:
        o, n, tr = self._slot[id(p)]
        view = self.data[o:o + n].view(p.shape)
        view.copy_(p.data)
        p.data = view
        if tr:
          p.grad = self.grad[o:o + n].view(p.shape)
        p._stk_flat = self

  def offset_of(self, p):
    o, _, tr = self._slot[id(p)]
    return o, tr

  def is_bound(self):
    """True while every parameter still aliases the flat buffers (``.to()`` or
    ``zero_grad(set_to_none=True)`` can break that; the owner then rebuilds)."""
    base = self.data.data_ptr()
    gbase = self.grad.data_ptr()
    for p in self.params:
      o, n, tr = self._slot[id(p)]
      if p.data.data_ptr() != base + 4 * o or p.device != self.device:
        return False
      if tr and (p.grad is None or p.grad.data_ptr() != gbase + 4 * o):
        return False
    return True

  def rebind_grads(self):
    for p in self.params:
      o, n, tr = self._slot[id(p)]
      if tr and (p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o):
        old = p.grad
        p.grad = self.grad[o:o + n].view(p.shape)
        if old is not None:
          with torch.no_grad():
            p.grad.copy_(old)
        else:
          p.grad.zero_()

  def trainable_views(self, flat):
    """Per-parameter views of another flat buffer with the same layout (Adam moments, EMA), in
    ``model.parameters()`` order (the order of the reference's ``shadow_params`` list)."""
    out = []
    for p in self.module_order:
      o, n, tr = self._slot[id(p)]
      if tr:
        out.append(flat[o:o + n].view(p.shape))
    return out

  def trainable_params(self):
    return [p for p in self.module_order if self._slot[id(p)][2]]


def flat_of(params):
  """The FlatParams shared by *all* of ``params`` (None if they are not flat-backed)."""
  owner = None
  for p in params:
    f = getattr(p, '_stk_flat', None)
    if f is None:
      return None
    if owner is None:
      owner = f
    elif owner is not f:
      return None
  return owner
