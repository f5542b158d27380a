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
import contextlib
import ctypes
import os
import warnings
import weakref

import numpy as np
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
from .graph import Conv, Graph, Runtime


class _WprepDesc(ctypes.Structure):        # StkWprepDesc of include/stk.h
  _fields_ = [('w', ctypes.c_void_p), ('wp', ctypes.c_void_p), ('sm', ctypes.c_long), ('sk', ctypes.c_long),
              ('M', ctypes.c_int), ('Kc', ctypes.c_int), ('Mpad', ctypes.c_int), ('taps', ctypes.c_int),
              ('flip', ctypes.c_int), ('reserved', ctypes.c_int)]


class _GnFoldDesc(ctypes.Structure):       # StkGnFoldDesc of include/stk.h
  _fields_ = [('part', ctypes.c_void_p), ('dgamma', ctypes.c_void_p), ('dbeta', ctypes.c_void_p),
              ('N', ctypes.c_int), ('C', ctypes.c_int)]


# STK_LIB_ONLY (default: on under STK_POISON, i.e. in the test suites): the engine's backward may launch kernels of libstk.so
# ONLY between its first op and the join of the side stream.  A gfx950 packed-fp32 instruction form returns wrong lanes beside
# another kernel's MFMAs (DESIGN.md "The hazard"): libstk.so is built without that form and checked by disassembly, torch's own
# kernels contain it (profiles/r04_torch_hip_pk_scan.json) -- so no torch kernel may run while the side stream carries weight
# gradients.  The guard turns that launch-order argument into a check: any ATen operator on a device tensor dispatched
# inside the window (a hook, a logging op, a future torch call inside an op) raises.
_LIB_ONLY = os.environ.get('STK_LIB_ONLY', '1' if _POISON else '0') == '1'
# STK_WP_SIDE=0: the data-gradient weight blocks are prepared in front of the forward with the forward blocks (A/B switch)
_WP_SIDE = os.environ.get('STK_WP_SIDE', '1') != '0'

from torch.utils._python_dispatch import TorchDispatchMode


class LibraryKernelsOnly(TorchDispatchMode):
  """Dispatch mode of the backward's launch window: records (record=True) or refuses every ATen operator that touches a
  device tensor, except the view / metadata operators that launch nothing."""
  NO_LAUNCH = frozenset(('view', '_unsafe_view', 'detach', 'alias', 'slice', 'select', 'as_strided', 'expand', 'permute',
                         'transpose', 't', 'unsqueeze', 'squeeze', 'narrow', 'unbind', 'split', 'reshape', '_reshape_alias',
                         'view_as', 'size', 'stride', 'sym_size', 'sym_stride', 'sym_numel', 'numel', 'dim', 'is_pinned',
                         'storage_offset', 'sym_storage_offset', 'empty', 'empty_like', 'empty_strided', 'new_empty',
                         'new_empty_strided', 'lift_fresh', 'is_contiguous', 'is_non_overlapping_and_dense',
                         'is_strides_like_format'))

  def __init__(self, record=False):
    super().__init__()
    self.record = record
    self.seen = []

  @staticmethod
  def _on_device(x):
    if isinstance(x, torch.Tensor):
      return x.is_cuda
    if isinstance(x, (list, tuple)):
      return any(LibraryKernelsOnly._on_device(v) for v in x)
    return False

  def __torch_dispatch__(self, func, types, args=(), kwargs=None):
    kwargs = kwargs or {}
    out = func(*args, **kwargs)
    name = getattr(getattr(func, 'overloadpacket', func), '__name__', str(func))
    ns = getattr(func, 'namespace', 'aten')
    if ns == 'aten' and name not in self.NO_LAUNCH and (self._on_device(args) or self._on_device(tuple(kwargs.values()))
                                                         or self._on_device(out)):
      self.seen.append(name)
      if not self.record:
        raise RuntimeError(f'torch operator aten::{name} launched on the device inside the engine\'s backward window: only '
                           f'libstk.so kernels may run beside the side stream\'s weight gradients (gfx950 packed-fp32 hazard, '
                           f'DESIGN.md "The hazard"; STK_LIB_ONLY=0 switches this check off)')
    return out


_LIB_ONLY_RECORDER = None     # tests: a LibraryKernelsOnly(record=True) to use instead of the raising one


@contextlib.contextmanager
def _launch_window(active):
  if not active or torch.cuda.is_current_stream_capturing():
    yield None
    return
  mode = _LIB_ONLY_RECORDER if _LIB_ONLY_RECORDER is not None else LibraryKernelsOnly()
  with mode:
    yield mode


_SIDE_STREAMS = {}     # (device index, launch stream handle) -> (side stream, overlap ratio, checked)

def _overlap_ratio(main, cand, cycles=400000):
  """Time of one spin kernel on each of the two streams, started together, over the time of one alone: ~1 when the streams run
  side by side, ~2 when they are served one after the other."""
  def timed(both):
    t0, t1, go = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event()
    t0.record(main)
    if both:
      go.record(main)
      cand.wait_event(go)
      with torch.cuda.stream(cand):
        torch.cuda._sleep(cycles)
      done = torch.cuda.Event()
      done.record(cand)
    with torch.cuda.stream(main):
      torch.cuda._sleep(cycles)
    if both:
      main.wait_event(done)
    t1.record(main)
    t1.synchronize()
    return t0.elapsed_time(t1)
  timed(False)
  alone = min(timed(False) for _ in range(2))
  return min(timed(True) for _ in range(2)) / max(alone, 1e-6)


def checked_side_stream(device):
  """A stream that really runs beside the device's current stream.  HIP serves its streams from a handful of hardware queues
  (4 by default) and torch hands out pooled streams round robin: every fourth one shares the current stream's queue, and a
  "second stream" on that queue runs strictly after the first -- measured on the 32x32 net 41.75 instead of 38.8 ms per step,
  on the 256x256 net 44.7 instead of 40.8 (profiles/r04_side_stream_queue.txt).  Candidates are tried with a pair of spin
  kernels until one overlaps.  The result is cached per (device, current stream): every engine of the process that launches
  from that stream shares it, and an engine driven from ANOTHER stream (a user's stream context) gets a side stream checked
  against that one.  An unchecked pick (STK_SIDE_CHECK=0, or asked for during a capture) is never cached as checked: the
  first call outside a capture replaces it."""
  dev = torch.device(device)
  index = dev.index if dev.index is not None else torch.cuda.current_device()
  main = torch.cuda.current_stream(dev)
  key = (index, main.cuda_stream)
  hit = _SIDE_STREAMS.get(key)
  capturing = torch.cuda.is_current_stream_capturing()
  want_check = os.environ.get('STK_SIDE_CHECK', '1') != '0'
  if hit is not None and (hit[2] or not want_check or capturing):
    return hit[0]
  check = want_check and not capturing
  best, best_ratio = (hit[0], 1e9) if (hit is not None and not check) else (None, 1e9)
  for _ in range(8 if check else (0 if best is not None else 1)):
    cand = torch.cuda.Stream(dev)
    ratio = _overlap_ratio(main, cand) if check else 0.0
    if ratio < best_ratio:
      best, best_ratio = cand, ratio
    if ratio < 1.5:
      break
  _SIDE_STREAMS[key] = (best, best_ratio, check)
  return best


class SideStream:
  """A second HIP stream for the weight gradients of a backward pass (STK_WGRAD_STREAM=0: off).  Works the same way in
  eager launches and inside a hipGraph capture, where the event pairs become the fork / join edges of the graph."""

  def __init__(self, device):
    self.device = device
    self.stream = checked_side_stream(device)
    self.last = None

  def begin(self):
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(self.device))
    self.stream.wait_event(ev)
    return self.stream.cuda_stream

  def end(self):
    ev = torch.cuda.Event()
    ev.record(self.stream)
    self.last = ev
    return ev

  def main_waits(self, ev):
    torch.cuda.current_stream(self.device).wait_event(ev)

  def join(self):
    if self.last is not None:
      self.main_waits(self.last)
      self.last = None


class Context:
  """Buffers of one forward call, kept until its backward has run."""
  __slots__ = ('prog', 'act', 'gact', 'rt', 'released', 'seed_t', 'graphs', 'uses', 'pl', '__weakref__')

  def __init__(self, prog):
    self.prog = prog
    self.act = _arena(prog.graph.act_size, prog.device)
    # planes arena (+ the data-gradient scratch behind it); poisoned with 0xff = fp16 NaN under STK_POISON
    g = prog.graph
    self.pl = None
    if g.pl_bytes + g.dypl_bytes > 0:
      self.pl = torch.empty(g.pl_bytes + g.dypl_bytes, dtype=torch.uint8, device=prog.device)
      if _POISON:
        self.pl.fill_(0xff)
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
    self.ws2 = None              # workspace of the side stream (weight gradients), allocated on first use
    self.free = []
    # prepared weights (include/stk.h): arena + device-resident descriptor table, built on first use.
    # Table order: all forward blocks, then all data-gradient blocks, so a no-grad call prepares a prefix.
    self.wp = None
    self.wp_table = None
    self.wp_counts = (0, 0)      # entries: forward only, forward + data gradient
    self.wp_items = 0
    self.wp_frozen = 0           # entries valid under Executor.frozen_weights()
    self._segments = {}          # bucket_elems -> backward segments for the overlapped gradient exchange
    # deferred GroupNorm parameter-gradient folds (include/stk.h stk_gn_param_grad_batch): partial-sum arena + table
    self.gnpart = None
    self.gn_table = None
    self.gn_maxc = 0
    self.wp_dgrad_ready = None   # event of the side-stream preparation of the data-gradient weight blocks (Executor._prepare_weights)

  def build_gn_folds(self, gparam_base):
    g = self.graph
    if not getattr(g, 'gn_folds', None):
      return
    self.gnpart = _arena(g.gnpart_size, self.device)
    base = self.gnpart.data_ptr()
    descs = []
    for part_off, dgamma_off, dbeta_off, n, ch in g.gn_folds:
      d = _GnFoldDesc()
      d.part = base + 4 * part_off
      d.dgamma = gparam_base + 4 * dgamma_off if dgamma_off is not None else None
      d.dbeta = gparam_base + 4 * dbeta_off if dbeta_off is not None else None
      d.N, d.C = n, ch
      self.gn_maxc = max(self.gn_maxc, d.C)
      descs.append(d)
    assert ctypes.sizeof(_GnFoldDesc) == 32
    raw = b''.join(bytes(d) for d in descs)
    self.gn_table = torch.from_numpy(np.frombuffer(raw, dtype=np.uint8).copy()).to(self.device)

  def backward_segments(self, n_train, bucket_elems):
    """Cut the backward launch sequence where buckets of the flat gradient buffer become final.

    Returns [(op_end, [(lo, hi), ...]), ...]: after the first `op_end` ops of the backward order have run, the listed
    ranges of the flat gradient buffer (float offsets) will not be written again in this backward.  Buckets are formed
    from the TOP of the buffer down: the layout is module order (engine/flat.py), the backward visits the modules in
    reverse, so high offsets finish first; the stacked time-embedding projections at the bottom finish last."""
    key = int(bucket_elems)
    segs = self._segments.get(key)
    if segs is not None:
      return segs
    from .graph import Tensor
    ops = list(reversed(self.graph.ops))
    last = {}                                   # flat offset of a parameter -> index of its last writer
    size = {}
    for i, op in enumerate(ops):
      for v in vars(op).values():
        if isinstance(v, Tensor) and v.space == 'param' and v.goff is not None:
          last[v.goff] = i
          size[v.goff] = v.numel
    # walk the parameters from the top of the buffer down, closing a bucket every `bucket_elems`
    offs = sorted(last, reverse=True)
    buckets, hi, ready = [], n_train, -1
    for k, off in enumerate(offs):
      ready = max(ready, last[off])
      if hi - off >= key or k == len(offs) - 1:
        lo = 0 if k == len(offs) - 1 else off
        buckets.append((lo, hi, ready))
        hi = lo
    if not offs:
      buckets = [(0, n_train, -1)]
    # a bucket is ready once its own AND all earlier (higher) buckets' writers ran: keep the cut points monotone
    segs, run = [], -1
    for lo, hi, ready in buckets:
      run = max(run, ready)
      if segs and segs[-1][0] == run + 1:
        segs[-1][1].append((lo, hi))
      else:
        segs.append((run + 1, [(lo, hi)]))
    if segs[-1][0] != len(ops):                 # ops after the last parameter writer (input-gradient tail)
      segs.append((len(ops), []))
    self._segments[key] = segs
    return segs

  def build_wp(self, lib, param_base):
    g = self.graph
    self.wp = torch.empty(max(g.wp_bytes, 256), dtype=torch.uint8, device=self.device)
    if _POISON:
      self.wp.fill_(0xff)          # bf16 0xffff = NaN: a block that is used without having been prepared shows up
    base = self.wp.data_ptr()
    assert base % 256 == 0
    descs, items = [], 0
    for direction in (0, 1):
      for op in g.ops:
        if not isinstance(op, Conv) or op.wp_off[direction] is None:
          continue
        d = _WprepDesc()
        n = lib.conv2d_wp_desc(direction, param_base + 4 * op.w.off, op.w_layout, op.C1 + op.C2, op.Cout, op.KH, op.KW,
                               base + op.wp_off[direction], ctypes.byref(d))
        if n <= 0:
          raise RuntimeError(f'stk_conv2d_wp_desc failed for a {op.C1 + op.C2}->{op.Cout} {op.KH}x{op.KW} layer (rc={n})')
        items = max(items, n)
        descs.append(d)
      if direction == 0:
        n_fwd = len(descs)
    self.wp_counts = (n_fwd, len(descs))
    self.wp_items = items
    if descs:
      raw = b''.join(bytes(d) for d in descs)
      self.wp_table = torch.from_numpy(np.frombuffer(raw, dtype=np.uint8).copy()).to(self.device)

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
    # grad mode is OFF inside autograd.Function.forward: say explicitly that a backward will follow
    out, c = ex.run_forward(x, emb_in, sigma, training, need_xgrad, with_backward=True, flat=ex.flat)
    ctx.ex, ctx.c, ctx.need_xgrad = ex, c, need_xgrad
    ex._awaiting.add(c)       # a backward of this evaluation may still come (dropped with the autograd graph otherwise)
    return out

  @staticmethod
  def backward(ctx, gout):
    if ctx.c.released:
      raise RuntimeError('score-network backward called twice on the same forward; the engine frees '
                         'activations after the first backward')
    # Does this backward pass want parameter gradients at all?  torch.autograd.grad(out, x) (the divergence / ELBO
    # estimators of likelihood.py) never reaches the anchor's accumulation node; .backward() does.  Without them the
    # weight / bias / affine gradients are neither computed nor added into p.grad -- as in the reference, where autograd
    # only walks the branches that were asked for.
    pg = True
    try:
      pg = bool(torch._C._will_engine_execute_node(ctx.ex._anchor_acc))
    except Exception:      # API absent, or the anchor itself is among autograd.grad's inputs
      pg = True
    gx = ctx.ex.run_backward(ctx.c, gout, param_grads=pg)
    return None, None, None, (gx if ctx.need_xgrad else None), None, None


class Executor:
  def __init__(self, model, backend=None):
    self.model = model
    self.lib = backend if backend is not None else stk_lib.load()
    self.flat = None
    self.programs = {}
    self._anchor = None
    self._anchor_acc = None
    self.profiler = None     # engine.profile.KernelTimer or None
    self.use_graphs = os.environ.get('STK_GRAPHS', '1') != '0'
    self.graph_replays = 0
    # STK_WP=0: every conv call prepares its own weights (the plain C-ABI calls) instead of one batched launch per
    # forward; a debugging switch, results are bit-identical
    self.use_wp = os.environ.get('STK_WP', '1') != '0'
    # STK_WGRAD_STREAM=0: everything on one stream.  Default: weight gradients and the shortcut convolutions' backward run
    # on a second stream beside the data-gradient chain (matrix-pipe-bound work beside the HBM-bound GroupNorm / planes
    # passes: -0.65 ... -0.85 ms per step), and the backward is launched eagerly -- a hipGraph with the same fork / join
    # structure runs its branches no faster than one stream (cross-stream edges cost 15-30 us each inside a graph), and the
    # host stays ~30 ms ahead of the GPU anyway.  STK_BWD_GRAPH=1 replays the backward as a hipGraph again.
    # Two kernels sharing a CU exposed a hazard of packed-fp32 VALU code (see csrc/Makefile, profiles/r03_side_stream_race.txt):
    # the library is built without it and tests/test_gpu_model.py::test_two_streams_are_deterministic keeps watch.
    self.use_side = os.environ.get('STK_WGRAD_STREAM', '1') != '0'
    self._side = None
    self.bwd_graphs = os.environ.get('STK_BWD_GRAPH', '0' if self.use_side else '1') != '0'

    self._frozen = 0
    # gradient exchange overlapped with the backward (engine/ddp.py): when set, run_backward cuts its launch sequence
    # into segments and calls grad_hook(lo, hi) as soon as a bucket [lo, hi) of the flat gradient buffer is final
    self.grad_hook = None
    self.grad_bucket_elems = 16 << 20
    # network evaluations made under grad mode whose backward has not run yet (weak: an evaluation whose autograd graph
    # is dropped without a backward disappears by itself).  A step that evaluates the network more than once per loss
    # (training.mixed, the reconstruction term: losses.py:134-164, 295-320) runs one engine backward per evaluation inside
    # ONE .backward(); buckets may only leave with the last of them.
    self._awaiting = weakref.WeakSet()

  # -- parameters ---------------------------------------------------------------------------------
  def set_backend(self, backend):
    self.lib = backend
    self.programs.clear()

  def ensure_flat(self):
    if self.flat is not None and self.flat.is_bound():      # every parameter, raw addresses only (~150 us)
      return self.flat
    params = list(self.model.parameters())
    device = params[0].device
    if self.flat is None or self.flat.device != device or not self._layout_ok(params):
      self.flat = FlatParams(params, device, groups=self.model._flat_groups())
      self.programs.clear()
      self._anchor = torch.zeros((), dtype=torch.float32, device=device, requires_grad=True)
      with torch.enable_grad():
        self._anchor_acc = self._anchor.view_as(self._anchor).grad_fn.next_functions[0][0]   # its AccumulateGrad node
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
      g = Graph(self.flat, self.lib)
      out = self.model._emit(g, B, H, W, need_xgrad)
      g.finalize(out, self.lib)
      prog = self.programs[key] = Program(g, self.flat.device)
    return prog

  # -- prepared weights ---------------------------------------------------------------------------
  def _prepare_weights(self, prog, need_dgrad):
    """One launch that splits / re-lays-out the weights of every conv layer of `prog` for the split kernels
    (forward blocks; data-gradient blocks too when a backward may follow).  Runs before every forward -- the
    weights may have changed since the last one -- except inside :meth:`frozen_weights`."""
    if not (self.use_wp and self.lib.is_device):
      return
    if prog.wp is None:
      prog.build_wp(self.lib, self.flat.data.data_ptr())
    n = prog.wp_counts[1 if need_dgrad else 0]
    if n == 0 or (self._frozen and prog.wp_frozen >= n):
      return
    n_fwd = prog.wp_counts[0]
    dev = self.flat.device
    if (need_dgrad and n > n_fwd > 0 and self.use_side and _WP_SIDE and not self._frozen and
        not torch.cuda.is_current_stream_capturing()):
      # the data-gradient blocks are first read by the backward: prepare them on the side stream, beside the forward
      # (an HBM-bound re-layout of 250 MB beside matrix-pipe-bound GEMMs) instead of in front of it; the backward waits
      # for the event (Program.wp_dgrad_ready) before its first launch
      if self._side is None or self._side.device != dev:
        self._side = SideStream(dev)
      self.lib.conv2d_wprep_batch(prog.wp_table.data_ptr(), n_fwd, prog.wp_items, stk_lib.stream_ptr(dev))
      s = self._side.begin()              # behind everything launched so far: the optimizer update of the last step
      self.lib.conv2d_wprep_batch(prog.wp_table.data_ptr() + ctypes.sizeof(_WprepDesc) * n_fwd, n - n_fwd, prog.wp_items, s)
      prog.wp_dgrad_ready = self._side.end()
      self._side.last = None              # not a join target of the backward's fork / join: the event is waited for explicitly
    else:
      if need_dgrad:
        # the data-gradient blocks are rewritten on the launch stream right here: a pending side-stream preparation is superseded.
        # A forward that prepares the forward blocks only (no_grad, frozen weights) must NOT drop the event: an earlier training
        # forward of this program may still be waiting for its backward, whose data gradients read those blocks
        prog.wp_dgrad_ready = None
      self.lib.conv2d_wprep_batch(prog.wp_table.data_ptr(), n, prog.wp_items, stk_lib.stream_ptr(dev))
    prog.wp_frozen = n if self._frozen else 0

  @contextlib.contextmanager
  def frozen_weights(self):
    """Promise that the parameters do not change inside the block (a sampling loop: thousands of network
    evaluations on fixed weights, sampling.py:365-433): the weights are prepared on the first evaluation only."""
    self._frozen += 1
    try:
      yield self
    finally:
      self._frozen -= 1
      if self._frozen == 0:
        for prog in self.programs.values():
          prog.wp_frozen = 0

  # -- execution ----------------------------------------------------------------------------------
  def _copy_in(self, c, key, value):
    t = c.prog.graph.inputs.get(key)
    if t is None:
      return
    c.act[t.off:t.off + t.numel].view(t.shape).copy_(value.detach().reshape(t.shape))

  def _graphs_on(self):
    return self.use_graphs and self.lib.is_device and self.profiler is None

  def _runtime(self, c, training, seed, seed_dev=None, with_backward=True):
    prog, flat = c.prog, self.flat
    rt = Runtime(self.lib, stk_lib.stream_ptr(flat.device), c.act.data_ptr(),
                 c.gact.data_ptr() if c.gact is not None else 0,
                 flat.data.data_ptr(), flat.grad.data_ptr(), prog.const.data_ptr(),
                 prog.ws.data_ptr(), prog.graph.ws_bytes, training, seed, seed_dev)
    rt.prof = self.profiler
    rt.with_backward = with_backward
    if self.use_wp and prog.wp is not None and prog.wp_table is not None:
      rt.wp = prog.wp.data_ptr()
    if c.pl is not None:
      rt.pl = c.pl.data_ptr()
      rt.dypl = rt.pl + prog.graph.pl_bytes
      if with_backward and self.use_side and self.lib.is_device and getattr(prog.graph, 'own_dypl', False):
        if self._side is None or self._side.device != flat.device:
          self._side = SideStream(flat.device)
        if prog.ws2 is None:
          prog.ws2 = _arena(prog.graph.ws_bytes // 4, prog.device)
        rt.side, rt.ws2 = self._side, prog.ws2.data_ptr()
    if with_backward and prog.gn_table is not None:
      rt.gnpart, rt.gn_table, rt.gn_maxc = prog.gnpart.data_ptr(), prog.gn_table.data_ptr(), prog.gn_maxc
    return rt

  def _replay(self, c, direction, training, span=None, with_backward=True, param_grads=True):
    """Replay (capturing on first use) the hipGraph of one direction of this context; `span` = (begin, end) restricts a
    backward graph to that slice of the backward op order (segments of the overlapped gradient exchange)."""
    key = (direction, training, with_backward, param_grads) if span is None else (direction, training, span)
    g = c.graphs.get(key)
    if g is None:
      ops = c.prog.graph.ops
      torch.cuda.synchronize()
      g = torch.cuda.CUDAGraph()
      try:
        # thread_local: calls made by OTHER threads while this one captures (e.g. the RCCL watchdog of a
        # multi-GPU run polling its events) must not invalidate the capture
        with torch.cuda.graph(g, capture_error_mode='thread_local'):
          rt = self._runtime(c, training, 0, c.seed_t.data_ptr(), with_backward)   # stream = the capture stream
          rt.param_grads = param_grads
          if direction == 'fwd':
            for op in ops:
              op.forward(rt)
          elif span is None:
            for op in reversed(ops):
              op.backward(rt)
            rt.flush_folds()
            rt.join_side()
          else:
            for op in list(reversed(ops))[span[0]:span[1]]:
              op.backward(rt)
            rt.flush_folds()
            rt.join_side()
      except Exception as e:   # capture is an optimisation: report, disable, run eagerly
        warnings.warn(f'hipGraph capture failed ({e!r}); continuing with eager launches')
        self.use_graphs = False
        return False
      c.graphs[key] = g
    g.replay()
    self.graph_replays += 1
    return True

  def run_forward(self, x, emb_in, sigma, training, need_xgrad, with_backward=False, flat=None):
    if flat is None:                      # apply() has just checked the layout: one walk over the ~570 parameters per call
      flat = self.ensure_flat()
    with stk_lib.device_guard(flat.device):
      return self._run_forward(x, emb_in, sigma, training, need_xgrad, with_backward, flat)

  def _run_forward(self, x, emb_in, sigma, training, need_xgrad, with_backward, flat):
    B, _, H, W = x.shape
    prog = self.program(B, H, W, need_xgrad)
    c = prog.acquire()
    g = prog.graph
    self._copy_in(c, 'x', x)
    self._copy_in(c, 'emb', emb_in)
    if sigma is not None:
      self._copy_in(c, 'sigma', sigma)
    self._prepare_weights(prog, with_backward)
    if with_backward and prog.gn_table is None and getattr(g, 'gn_folds', None):
      prog.build_gn_folds(flat.grad.data_ptr())          # allocation + upload: never inside a hipGraph capture
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
      done = self._replay(c, 'fwd', training, with_backward=with_backward)
      if done:
        c.rt = self._runtime(c, training, 0, c.seed_t.data_ptr(), with_backward)
    if not done:
      rt = self._runtime(c, training, seed, None, with_backward)
      for op in g.ops:
        op.forward(rt)
      c.rt = rt
    o = g.output
    out = c.act[o.off:o.off + o.numel].view(o.shape).clone()
    return out, c

  def run_backward(self, c, gout, param_grads=True):
    with stk_lib.device_guard(self.flat.device):
      return self._run_backward(c, gout, param_grads)

  def _run_backward(self, c, gout, param_grads):
    prog, g, rt = c.prog, c.prog.graph, c.rt
    rt.param_grads = param_grads
    flat = self.flat
    self._awaiting.discard(c)
    self._last_ctx = weakref.ref(c)  # dynamic_range_report reads its gradient arena (valid until the context's next backward); weak: the report must not keep a released context alive
    if c.gact is None:
      c.gact = _arena(g.gact_size, prog.device)
    o = g.output
    c.gact[o.goff:o.goff + o.numel].view(o.shape).copy_(gout)
    if prog.wp_dgrad_ready is not None:                 # data-gradient weight blocks prepared on the side stream
      torch.cuda.current_stream(flat.device).wait_event(prog.wp_dgrad_ready)
    done = False
    hook = self.grad_hook if param_grads else None      # nothing to exchange after an input-gradient-only backward
    if hook is not None and len(self._awaiting) > 0:
      # another evaluation of this step still owes its backward (it accumulates into the same flat buffer): a bucket
      # handed over now would be reduced before that contribution arrives.  Plain backward; the LAST pending one hands
      # the buckets over (or OverlappedExchange.finish() reduces what nobody handed over).
      hook = None
    if hook is not None:
      # overlapped exchange: segment by segment, handing finished buckets to the hook (which starts their all-reduce
      # on the communicator's stream, ordered behind the launches made so far)
      segs = prog.backward_segments(flat.n_train, self.grad_bucket_elems)
      graphs = self._graphs_on() and rt.seed_dev is not None and self.bwd_graphs
      if not graphs:
        rt.gbase['act'] = c.gact.data_ptr()
        rt.gbase['param'] = flat.grad.data_ptr()
        rt.stream = stk_lib.stream_ptr(flat.device)
        rt.prof = self.profiler
      order = list(reversed(g.ops))
      begin = 0
      for end, ranges in segs:
        if end > begin and any(op.bwd_launches() for op in order[begin:end]):      # a span without launches is skipped
          if not (graphs and self._replay(c, 'bwd', rt.training, (begin, end))):
            if graphs:            # capture failed half way: finish eagerly with a consistent runtime
              graphs = False
              rt.gbase['act'] = c.gact.data_ptr()
              rt.gbase['param'] = flat.grad.data_ptr()
              rt.stream = stk_lib.stream_ptr(flat.device)
            last = end >= len(order)
            with _launch_window(_LIB_ONLY and rt.side is not None):
              for op in order[begin:end]:
                op.backward(rt)
              rt.flush_folds()
              if last or rt.side is None:
                rt.join_side()
            if not last and rt.side is not None and ranges:
              # A bucket is final once the main chain AND the side stream's weight gradients launched so far are done.  Joining the side
              # stream into the main one at every segment end made the main chain wait for it four times per step (with one weight-
              # gradient workgroup per CU the side stream runs further behind: exchange proxy +0.5 -> +1.5 ms per step); instead a third
              # stream waits for both and the bucket's all-reduce is issued from THAT stream's context -- the communicator's stream
              # then waits for it, the main chain for nobody.
              self._hand_over(rt, hook, ranges)
              begin = end
              continue
        begin = end
        for lo, hi in ranges:
          hook(lo, hi)
      if not graphs:
        rt.join_side()              # (no-op after a joined last segment; a skipped one must not leave side work unjoined)
      done = True
    elif self._graphs_on() and rt.seed_dev is not None and (self.bwd_graphs or not param_grads or rt.side is None):
      done = self._replay(c, 'bwd', rt.training, param_grads=param_grads)
    if not done:
      rt.gbase['act'] = c.gact.data_ptr()
      rt.gbase['param'] = flat.grad.data_ptr()
      rt.stream = stk_lib.stream_ptr(flat.device)
      rt.prof = self.profiler
      with _launch_window(_LIB_ONLY and rt.side is not None):
        for op in reversed(g.ops):
          op.backward(rt)
        rt.flush_folds()
        rt.join_side()
    gx = None
    xin = g.inputs['x']
    if xin.needs_grad:
      gx = c.gact[xin.goff:xin.goff + xin.numel].view(xin.shape).clone()
    prog.release(c)
    return gx

  def _hand_over(self, rt, hook, ranges):
    """Issue the all-reduce of finished buckets behind BOTH streams of the backward without making either wait for the other."""
    dev = self.flat.device
    main = torch.cuda.current_stream(dev)
    x = getattr(self, '_xchg_stream', None)
    if x is None or x.device != dev:
      x = self._xchg_stream = torch.cuda.Stream(dev)        # carries only event waits and the collectives' issue point
    ev = torch.cuda.Event()
    ev.record(main)
    x.wait_event(ev)
    if rt.side.last is not None:
      x.wait_event(rt.side.last)
    with torch.cuda.stream(x):
      for lo, hi in ranges:
        hook(lo, hi)

  def dynamic_range_report(self):
    """Per-image dynamic range of the output gradients of the LAST backward: the split convolutions scale a tensor by ONE power
    of two (include/stk.h "Planes"), which keeps every image at fp32 accuracy only while its own maximum lies within about five
    decades of the batch maximum (tests/test_gpu_kernels.py::test_conv_split_per_image_accuracy: 2e-7 down to 1e-4 of the
    maximum, 9e-7 at 1e-5, 8e-6 at 1e-6).  Reads the fp32 output gradients the backward left in the gradient arena (they live
    until the context's next backward) -- a diagnostic for every few hundred steps, not a by-product of the kernels: per-image
    maxima by atomic maximum made the 32 workgroups of an image hit one address at once and cost the GroupNorm backward 20-60 %.
    Returns [(layer, max over images, smallest non-zero image maximum, decades between them)] for the 3x3 convolutions whose
    output gradient exists as fp32, worst first.  Synchronises."""
    ref = getattr(self, '_last_ctx', None)
    c = ref() if ref is not None else None
    if c is None or c.gact is None:
      return []
    g = c.prog.graph
    names, his, los = [], [], []
    with torch.no_grad():
      for op in g.ops:
        if not isinstance(op, Conv) or op.KH != 3 or not (op.pl_dgrad or op.pl_wgrad):
          continue
        t = op.y
        if not t.needs_grad or t.goff is None or op.N < 2:
          continue
        per = c.gact[t.goff:t.goff + t.numel].view(op.N, -1).abs().amax(dim=1)
        ok = torch.isfinite(per) & (per > 0)
        # hi / lo stay on the device: ONE transfer for all layers below (a .max() / .min() pair per layer was two host syncs
        # per convolution)
        his.append(torch.where(ok, per, torch.zeros_like(per)).max())
        los.append(torch.where(ok, per, torch.full_like(per, float('inf'))).min())
        names.append((t.name, ok.sum()))
      if not names:
        return []
      table = torch.stack([torch.stack(his), torch.stack(los), torch.stack([n for _, n in names]).to(his[0].dtype)]).cpu()
    rows = []
    for i, (name, _) in enumerate(names):
      hi, lo, n = float(table[0, i]), float(table[1, i]), int(table[2, i])
      if n >= 2 and lo > 0 and np.isfinite(lo):
        rows.append((name, hi, lo, float(np.log10(hi / lo))))
    rows.sort(key=lambda r: -r[3])
    return rows

  def apply(self, x, emb_in, sigma=None):
    """Differentiable network evaluation (forward now, backward when autograd asks)."""
    self.ensure_flat()
    training = self.model.training
    x = x.contiguous()
    if torch.is_grad_enabled():
      return _NetFn.apply(self, training, self._anchor, x, emb_in, sigma)
    out, c = self.run_forward(x, emb_in, sigma, training, False, flat=self.flat)
    c.prog.release(c)
    return out
