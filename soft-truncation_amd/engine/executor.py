"""Runs a planned score-network graph on the C-ABI kernels and bridges it into torch.autograd.

One :class:`Executor` per model.  ``Executor.apply(x, emb_in, sigma)`` behaves like the
reference's ``NCSNpp.forward`` body (models/ncsnpp.py:258-432) from autograd's point of view
-- one differentiable node -- while inside it is a fixed launch sequence over pre-planned
HBM buffers with hand-written backward kernels (engine/graph.py).

hipGraphs: because the plan is static (fixed buffers, fixed launch order, no allocation, no
host sync), the forward and the backward launch sequences of a context are captured once into
hipGraphs and replayed afterwards -- ~650 launches per direction become one graph launch, which
removes the host launch gaps between the ~20-500 us kernels (the reference would need a tracing
compiler for this; here it falls out of the design).  The only per-call value inside the
network, the dropout seed, is read from device memory (``seed_dev`` of ``stk_gn_*``), so a
replay draws fresh masks.  ``STK_GRAPHS=0`` disables capture; attaching a kernel timer
(events cannot be recorded inside a replay) does so too.

Backend selection is explicit and never silent: the default backend is the HIP library
(``engine.lib.load()``, raises if it is not built); a test may inject another implementation
of include/stk.h (the oracle's CPU restatement) with ``set_backend``.
"""
import os
import warnings

import torch

from . import lib as stk_lib

# STK_POISON=1 (debugging aid): fill every arena with NaN when it is allocated, so that an op reading memory that no
# op has written shows up as NaN in the output instead of depending on what the allocator hands back.
_POISON = os.environ.get('STK_POISON', '0') == '1'


def _arena(n, device):
  t = torch.empty(max(n, 1), dtype=torch.float32, device=device)
  if _POISON:
    t.fill_(float('nan'))
  return t
from .flat import FlatParams
from .graph import Graph, Runtime


class Context:
  """Buffers of one forward call, kept until its backward has run."""
  __slots__ = ('prog', 'act', 'gact', 'rt', 'released', 'seed_t', 'graphs', 'uses')

  def __init__(self, prog):
    self.prog = prog
    self.act = _arena(prog.graph.act_size, prog.device)
    self.gact = None
    self.rt = None
    self.released = False
    self.seed_t = None        # device int64 holding the per-call dropout seed (graph mode)
    self.graphs = {}          # ('fwd'|'bwd', training) -> torch.cuda.CUDAGraph
    self.uses = 0


class Program:
  """A finalized graph for one input signature plus its constant pool and context pool."""

  def __init__(self, graph, device):
    self.graph = graph
    self.device = device
    const = torch.zeros(max(graph.const_size, 1), dtype=torch.float32)
    for off, arr in graph.const_chunks:
      const[off:off + arr.size] = torch.from_numpy(arr)
    self.const = const.to(device)
    self.ws = _arena(graph.ws_bytes // 4, device)
    self.free = []

  def acquire(self):
    c = self.free.pop() if self.free else Context(self)
    c.released = False
    return c

  def release(self, c):
    if not c.released:
      c.released = True
      c.rt = None
      if len(self.free) < 4:
        self.free.append(c)


class _NetFn(torch.autograd.Function):
  @staticmethod
  def forward(ctx, ex, training, anchor, x, emb_in, sigma):
    need_xgrad = bool(ctx.needs_input_grad[3])
    out, c = ex.run_forward(x, emb_in, sigma, training, need_xgrad)
    ctx.ex, ctx.c, ctx.need_xgrad = ex, c, need_xgrad
    return out

  @staticmethod
  def backward(ctx, gout):
    if ctx.c.released:
      raise RuntimeError('score-network backward called twice on the same forward; the engine frees '
                         'activations after the first backward')
    gx = ctx.ex.run_backward(ctx.c, gout)
    return None, None, None, (gx if ctx.need_xgrad else None), None, None


class Executor:
  def __init__(self, model, backend=None):
    self.model = model
    self.lib = backend if backend is not None else stk_lib.load()
    self.flat = None
    self.programs = {}
    self._anchor = None
    self.profiler = None     # engine.profile.KernelTimer or None
    self.use_graphs = os.environ.get('STK_GRAPHS', '1') != '0'
    self.graph_replays = 0

  # -- parameters ---------------------------------------------------------------------------------
  def set_backend(self, backend):
    self.lib = backend
    self.programs.clear()

  def ensure_flat(self):
    params = list(self.model.parameters())
    device = params[0].device
    if self.flat is None or self.flat.device != device or not self._layout_ok(params):
      self.flat = FlatParams(params, device, groups=self.model._flat_groups())
      self.programs.clear()
      self._anchor = torch.zeros((), dtype=torch.float32, device=device, requires_grad=True)
    else:
      self.flat.rebind_grads()
    if self.lib.is_device != (device.type == 'cuda'):
      raise RuntimeError(f'backend {self.lib.backend} cannot run a model on {device}: the score network '
                         f'runs on the HIP kernels only (no CPU / PyTorch fallback)')
    return self.flat

  def _layout_ok(self, params):
    base = self.flat.data.data_ptr()
    for p in params:
      slot = self.flat._slot.get(id(p))
      if slot is None or p.data.data_ptr() != base + 4 * slot[0]:
        return False
    return True

  # -- programs -----------------------------------------------------------------------------------
  def program(self, B, H, W, need_xgrad):
    key = (B, H, W, need_xgrad)
    prog = self.programs.get(key)
    if prog is None:
      g = Graph(self.flat)
      out = self.model._emit(g, B, H, W, need_xgrad)
      g.finalize(out, self.lib)
      prog = self.programs[key] = Program(g, self.flat.device)
    return prog

  # -- execution ----------------------------------------------------------------------------------
  def _copy_in(self, c, key, value):
    t = c.prog.graph.inputs.get(key)
    if t is None:
      return
    c.act[t.off:t.off + t.numel].view(t.shape).copy_(value.detach().reshape(t.shape))

  def _graphs_on(self):
    return self.use_graphs and self.lib.is_device and self.profiler is None

  def _runtime(self, c, training, seed, seed_dev=None):
    prog, flat = c.prog, self.flat
    rt = Runtime(self.lib, stk_lib.stream_ptr(flat.device), c.act.data_ptr(),
                 c.gact.data_ptr() if c.gact is not None else 0,
                 flat.data.data_ptr(), flat.grad.data_ptr(), prog.const.data_ptr(),
                 prog.ws.data_ptr(), prog.graph.ws_bytes, training, seed, seed_dev)
    rt.prof = self.profiler
    return rt

  def _replay(self, c, direction, training):
    """Replay (capturing on first use) the hipGraph of one direction of this context."""
    key = (direction, training)
    g = c.graphs.get(key)
    if g is None:
      ops = c.prog.graph.ops
      torch.cuda.synchronize()
      g = torch.cuda.CUDAGraph()
      try:
        # thread_local: calls made by OTHER threads while this one captures (e.g. the RCCL watchdog of a
        # multi-GPU run polling its events) must not invalidate the capture
        with torch.cuda.graph(g, capture_error_mode='thread_local'):
          rt = self._runtime(c, training, 0, c.seed_t.data_ptr())   # stream = the capture stream
          if direction == 'fwd':
            for op in ops:
              op.forward(rt)
          else:
            for op in reversed(ops):
              op.backward(rt)
      except Exception as e:   # capture is an optimisation: report, disable, run eagerly
        warnings.warn(f'hipGraph capture failed ({e!r}); continuing with eager launches')
        self.use_graphs = False
        return False
      c.graphs[key] = g
    g.replay()
    self.graph_replays += 1
    return True

  def run_forward(self, x, emb_in, sigma, training, need_xgrad):
    flat = self.ensure_flat()
    B, _, H, W = x.shape
    prog = self.program(B, H, W, need_xgrad)
    c = prog.acquire()
    g = prog.graph
    self._copy_in(c, 'x', x)
    self._copy_in(c, 'emb', emb_in)
    if sigma is not None:
      self._copy_in(c, 'sigma', sigma)
    seed = 0
    if training and self.model._uses_dropout():
      seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    c.uses += 1
    done = False
    if self._graphs_on() and c.uses > 1:          # first use of a context runs eagerly (warm-up)
      if c.seed_t is None:
        c.seed_t = torch.zeros(1, dtype=torch.int64, device=flat.device)
      if c.gact is None and torch.is_grad_enabled():
        c.gact = _arena(g.gact_size, prog.device)
      c.seed_t.fill_(seed)
      done = self._replay(c, 'fwd', training)
      if done:
        c.rt = self._runtime(c, training, 0, c.seed_t.data_ptr())
    if not done:
      rt = self._runtime(c, training, seed)
      for op in g.ops:
        op.forward(rt)
      c.rt = rt
    o = g.output
    out = c.act[o.off:o.off + o.numel].view(o.shape).clone()
    return out, c

  def run_backward(self, c, gout):
    prog, g, rt = c.prog, c.prog.graph, c.rt
    flat = self.flat
    if c.gact is None:
      c.gact = _arena(g.gact_size, prog.device)
    o = g.output
    c.gact[o.goff:o.goff + o.numel].view(o.shape).copy_(gout)
    done = False
    if self._graphs_on() and rt.seed_dev is not None:
      done = self._replay(c, 'bwd', rt.training)
    if not done:
      rt.gbase['act'] = c.gact.data_ptr()
      rt.gbase['param'] = flat.grad.data_ptr()
      rt.stream = stk_lib.stream_ptr(flat.device)
      rt.prof = self.profiler
      for op in reversed(g.ops):
        op.backward(rt)
    gx = None
    xin = g.inputs['x']
    if xin.needs_grad:
      gx = c.gact[xin.goff:xin.goff + xin.numel].view(xin.shape).clone()
    prog.release(c)
    return gx

  def apply(self, x, emb_in, sigma=None):
    """Differentiable network evaluation (forward now, backward when autograd asks)."""
    self.ensure_flat()
    training = self.model.training
    x = x.contiguous()
    if torch.is_grad_enabled():
      return _NetFn.apply(self, training, self._anchor, x, emb_in, sigma)
    out, c = self.run_forward(x, emb_in, sigma, training, False)
    c.prog.release(c)
    return out
