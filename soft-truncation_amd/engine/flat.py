"""Flat parameter / gradient storage.

All trainable parameters of the score network live in ONE contiguous fp32 buffer (and their
gradients in a second one) laid out for the device rather than for the module tree:

* the fused clip + Adam + EMA kernels stream the whole model in three launches instead of the
  reference's Python loops over 564 tensors (losses.py:55-56, models/ema.py:50-51);
* the RCCL gradient all-reduce works on a handful of large contiguous buckets;
* the per-block time-embedding projections (46 x nn.Linear(512, Cout), models/layerspp.py:240)
  are placed back to back so the engine evaluates them as one [B,512]x[512,sum Cout] GEMM.

* the q / k / v projections of an attention block (three NIN layers with [in, out] weights, models/layerspp.py:81-83)
  are interleaved column-wise into one [in, 3 out] matrix, so the engine evaluates them -- and their data and weight
  gradients -- as ONE 1x1 convolution with 3 C output channels (a ``('cols', [W0, W1, W2])`` group: the three
  parameters are strided views of the block).

``nn.Parameter`` objects stay what the reference's code expects (``state_dict`` keys, shapes,
``.grad``) -- their ``.data`` / ``.grad`` are views into the flat buffers.
"""
import torch

ALIGN = 64  # floats


def _round_up(n, a=ALIGN):
  return (n + a - 1) // a * a


class FlatParams:
  def __init__(self, params, device, groups=()):
    """params: ordered list of nn.Parameter.  groups: each either a list of parameters that must be laid out
    contiguously and unpadded, in order (e.g. all Dense_0 weights), or ``('cols', [W0, W1, ...])``: 2-D parameters
    of one shape [R, Cn] interleaved as the column blocks of one [R, K Cn] matrix."""
    self.device = torch.device(device)
    params = list(params)
    self.module_order = params      # the order of model.parameters(): what state_dicts / EMA lists use
    seen = set()
    order = []
    self.group_ranges = []
    self._strides = {}              # id(p) -> strides of a parameter that is not a contiguous slice (cols groups)
    self._cols = {}                 # id(first member) -> (float offset of the block, [members])
    for grp in groups:
      cols = isinstance(grp, tuple) and len(grp) == 2 and grp[0] == 'cols'
      grp = [p for p in (grp[1] if cols else grp)]
      for p in grp:
        assert id(p) not in seen
        seen.add(id(p))
      if cols:
        assert all(p.dim() == 2 and p.shape == grp[0].shape and p.requires_grad == grp[0].requires_grad for p in grp)
      order.append((grp, cols))
    for p in params:
      if id(p) not in seen:
        seen.add(id(p))
        order.append(([p], False))
    # trainable first, frozen after (EMA / Adam / all-reduce cover [0, n_train))
    trainable = [g for g in order if g[0][0].requires_grad]
    frozen = [g for g in order if not g[0][0].requires_grad]
    self._slot = {}
    off = 0
    self.params = []

    def place(grp, cols, off, tr):
      if cols:
        R, Cn = grp[0].shape
        K = len(grp)
        self._cols[id(grp[0])] = (off, grp)
        for i, p in enumerate(grp):
          self._slot[id(p)] = (off + i * Cn, p.numel(), tr)
          self._strides[id(p)] = (K * Cn, 1)
          self.params.append(p)
        return off + R * K * Cn
      for p in grp:
        self._slot[id(p)] = (off, p.numel(), tr)
        self.params.append(p)
        off += p.numel()
      return off

    for grp, cols in trainable:
      start = off
      off = place(grp, cols, off, True)
      self.group_ranges.append((start, off))
      off = _round_up(off)
    self.n_train = off
    for grp, cols in frozen:
      off = _round_up(place(grp, cols, off, False))
    self.n_total = off
    self.data = torch.zeros(self.n_total, dtype=torch.float32, device=self.device)
    self.grad = torch.zeros(max(self.n_train, 1), dtype=torch.float32, device=self.device)
    with torch.no_grad():
      for p in self.params:
        tr = self._slot[id(p)][2]
        view = self.view_of(self.data, p)
        view.copy_(p.data)
        p.data = view
        if tr:
          p.grad = self.view_of(self.grad, p)
        p._stk_flat = self
    self._snapshot_ptrs()

  def view_of(self, buf, p):
    """The view of `buf` (a buffer with this layout) that belongs to parameter `p`."""
    o, n, _ = self._slot[id(p)]
    st = self._strides.get(id(p))
    if st is None:
      return buf[o:o + n].view(p.shape)
    return torch.as_strided(buf, tuple(p.shape), st, o)

  def cols_block(self, members):
    """Float offset of the [R, K Cn] block that interleaves exactly `members` (in this order), or None."""
    ent = self._cols.get(id(members[0]))
    if ent is None or len(ent[1]) != len(members) or any(a is not b for a, b in zip(ent[1], members)):
      return None
    return ent[0]

  def offset_of(self, p):
    o, _, tr = self._slot[id(p)]
    return o, tr

  def is_bound(self):
    """True while EVERY parameter still aliases the flat buffers (``.to()``, ``load_state_dict`` onto new storage, a
    single ``p.data = ...`` or ``p.grad = None`` -- all legal against the reference -- break that; the owner then
    rebuilds / rebinds).  Called on every network evaluation and every ``zero_grad``: the walk compares raw addresses
    against two lists made at construction time (570 parameters: ~50 us for the data pointers, ~100 us for the gradients;
    the ~4 ms per step measured in round 2 came from ``list(model.parameters())`` and ``p.data`` aliases, not from this)."""
    if [p.data_ptr() for p in self.params] != self._data_ptrs:
      return False
    return self.grads_bound()

  def grads_bound(self):
    for p, want in self._grad_ptrs:
      g = p.grad
      if g is None or g.data_ptr() != want:
        return False
    return True

  def _snapshot_ptrs(self):
    base, gbase = self.data.data_ptr(), self.grad.data_ptr()
    self._data_ptrs = [base + 4 * self._slot[id(p)][0] for p in self.params]
    self._grad_ptrs = [(p, gbase + 4 * self._slot[id(p)][0]) for p in self.params if self._slot[id(p)][2]]

  def rebind_grads(self):
    for p in self.params:
      o, n, tr = self._slot[id(p)]
      if tr and (p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o):
        old = p.grad
        p.grad = self.view_of(self.grad, p)
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
        out.append(self.view_of(flat, p))
    return out

  def trainable_params(self):
    return [p for p in self.module_order if self._slot[id(p)][2]]


def flat_of(params, full=True):
  """The FlatParams shared by *all* of ``params`` (None if they are not flat-backed, or if the list mixes in parameters
  of another buffer / plain tensors -- the fused optimizer and the flat EMA update whole flat ranges and would silently
  ignore those).  ``full=False`` (per-step callers that pass the same list every time): first / middle / last stand for
  all, plus the length."""
  owner = None
  params = params if isinstance(params, (list, tuple)) else list(params)
  probe = params
  if not full and len(params) > 3:
    probe = (params[0], params[len(params) // 2], params[-1])
  for p in probe:
    f = getattr(p, '_stk_flat', None)
    if f is None:
      return None
    if owner is None:
      owner = f
    elif owner is not f:
      return None
  if owner is not None:
    if full:
      if any(id(p) not in owner._slot for p in params):      # a stale tag: the parameter belongs to an older layout
        return None
    if len(params) != len(owner.params) and len(params) != len(owner._grad_ptrs):
      # neither the model's parameters nor its trainable ones: a partial list cannot be stepped as one flat range
      return None
  return owner
