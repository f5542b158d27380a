"""Static op graph of one score-network evaluation and its hand-written backward.

The reference runs the U-Net as ~37.7k eager ATen calls per training step with the autograd
tape rebuilt every step.  Here the network is lowered ONCE per (batch, height,
width) into a flat list of ops over pre-planned buffers:

* every activation lives at a fixed offset of one HBM arena (288 GB per GPU makes "keep
  everything resident" the cheap choice -- no allocator traffic, no recomputation);
* every op launches C-ABI kernels (include/stk.h) on the current HIP stream, forward and
  backward; the backward list is the forward list reversed, with gradient accumulation
  (beta = 0 for the first writer of a gradient buffer, 1 afterwards) resolved at plan time;
* parameter gradients are accumulated straight into the flat gradient buffer that the fused
  optimizer and the RCCL all-reduce consume (engine/flat.py, engine/optim.py, engine/ddp.py).

Because the plan is static, a whole forward/backward is a fixed launch sequence that can be
captured into a hipGraph (engine/executor.py).
"""
import math
import os

import numpy as np

SQRT2 = float(np.float32(np.sqrt(2.)))
# What runs on the side stream (engine/executor.SideStream) is fixed since round 5: every weight gradient, nothing else -- the shortcut
# convolutions' backward and the forward's shortcuts were measured there twice and lost (profiles/r05_insitu_sweeps.txt,
# profiles/r06_insitu_sweeps.txt); their switches are retired.
# Workgroups of a weight gradient launched on the MAIN stream (one-stream mode, the profiler's eager steps), 0 = the same count as on the side
# stream.  512 (two per CU) is the faster setting for a kernel that has the chip to itself, but a different K split sums its slabs in a
# different order, and the one-stream backward is the bit-for-bit reference of the two-stream one (tests/test_gpu_model.py::
# test_two_streams_are_deterministic): the default keeps the arithmetic of the two modes identical.
_X2W_WGS_ALONE = int(os.environ.get('STK_X2W_WGS_ALONE', '0'))
_SIDE_DELAY = 0             # tests: spin cycles that hold the side stream back before a launch (tests/_model_cases.py)
_SIDE_DELAY_FILTER = None
_ALIGN = 64  # floats (256 B) -- keeps every buffer float4-aligned
# A convolution's record buffer: three 256-float scale records -- |x1|, |x2|, |dy| (include/stk.h "amax")
AMAX = 768


def _round_up(n, a=_ALIGN):
  return (n + a - 1) // a * a


class Tensor:
  """Symbolic fp32 tensor: a shape plus an offset into one of the runtime's address spaces."""
  __slots__ = ('shape', 'numel', 'space', 'off', 'goff', 'needs_grad', 'name', 'seen', 'external_grad',
               'producer', 'pl_off', 'pl_rec', 'pl_maker', 'f32_fwd', 'f32_bwd')

  def __init__(self, shape, space, off, needs_grad, name):
    self.shape = tuple(int(s) for s in shape)
    self.numel = int(np.prod(self.shape)) if len(self.shape) else 1
    self.space = space          # 'act' | 'param' | 'const'
    self.off = off              # float offset in its space
    self.goff = None            # float offset of the gradient (act: grad arena, param: flat grad)
    self.needs_grad = needs_grad
    self.name = name
    self.seen = 0               # number of gradient writers already planned (backward order)
    self.external_grad = False  # gradient is seeded from outside (network output)
    self.producer = None        # the op whose output this is (Graph.add)
    # planes (include/stk.h "Planes"): byte offset of the tensor's pre-split copy in the planes arena, the (tensor,
    # float offset) of its 256-float scale record, and the op that writes them in the forward pass
    self.pl_off = None
    self.pl_rec = None
    self.pl_maker = None
    # does anybody read the fp32 copy of this tensor in the forward pass / in a backward pass?  (Graph.finalize; a
    # GroupNorm whose consumers all take planes skips writing it)
    self.f32_fwd = True
    self.f32_bwd = True

  def __repr__(self):
    return f'T({self.name}{list(self.shape)}@{self.space}+{self.off})'


class Runtime:
  """Per-call view of the buffers: base addresses, stream, library, training flag."""

  def __init__(self, lib, stream, act, gact, param, gparam, const, ws, ws_bytes, training, seed,
               seed_dev=None):
    self.lib = lib
    self.stream = stream
    self.base = {'act': act, 'param': param, 'const': const}
    self.gbase = {'act': gact, 'param': gparam}
    self.ws = ws
    self.ws_bytes = ws_bytes
    self.training = training
    self.seed = seed
    self.seed_dev = seed_dev
    self.wp = 0           # base address of the program's prepared-weight arena (0: every conv prepares per call)
    self.with_backward = True   # False: a no-grad evaluation (nothing is kept for a backward pass)
    self.pl = 0           # base address of the context's planes arena (pre-split conv operands)
    self.dypl = 0         # ... and of its scratch for the planes of the gradient a data-gradient call consumes
    self.prof = None      # optional engine.profile.KernelTimer: HIP events around the contraction launches
    # deferred GroupNorm parameter-gradient folds (stk_gn_param_grad_batch): base of the program's partial-sum arena,
    # device address of its descriptor table (entries in backward order), and the pending range of entries
    # False: the caller asked for the INPUT gradient only (torch.autograd.grad(out, x): likelihood.get_div_fn / get_elbo_fn)
    # -- g() of a parameter is None, so no weight / bias / affine gradient is computed and flat.grad is not touched
    self.param_grads = True
    self.gnpart = 0
    self.gn_table = 0
    self.gn_maxc = 0
    self._fold_lo = None
    self._fold_hi = None
    # weight gradients on a second stream (engine/executor.SideStream; None = everything on `stream`): they are
    # matrix-pipe-bound and independent of the data-gradient chain, whose GroupNorm backward / planes passes are HBM-bound.
    # `ws2` = the side stream's own workspace.  Every convolution then owns the planes of its output gradient
    # (Conv.dypl_off) instead of sharing one scratch buffer: a shared buffer would make the main chain wait for the side
    # stream before every split pass, and inside a hipGraph each such cross-stream edge costs 15-30 us.
    self.side = None
    self.ws2 = 0
    self._cur = None

  def side_launch(self, fn, *args):
    """fn(*args, side stream), ordered behind everything launched on the main stream so far."""
    s = self.side.begin()
    if _SIDE_DELAY and (_SIDE_DELAY_FILTER is None or _SIDE_DELAY_FILTER(self._cur)):   # debugging: hold the side stream back
      import torch
      with torch.cuda.stream(self.side.stream):
        torch.cuda._sleep(_SIDE_DELAY)
    fn(*args, s)
    self.side.end()

  def join_side(self):
    if self.side is not None:
      self.side.join()

  def defer_fold(self, index):
    """A GroupNorm backward left its per-(sample, channel) sums behind; entry `index` of the table folds them."""
    if self._fold_hi is not None and index != self._fold_hi:      # a skipped entry (frozen parameters): close the run
      assert index > self._fold_hi, 'folds must be deferred in table order'
      self.flush_folds()
    if self._fold_lo is None:
      self._fold_lo = index
    self._fold_hi = index + 1

  def flush_folds(self):
    """One launch for every fold deferred since the last flush (end of a backward segment / of the backward)."""
    if self._fold_lo is None:
      return
    lo, hi = self._fold_lo, self._fold_hi
    self._fold_lo = self._fold_hi = None
    self.lib.gn_param_grad_batch(self.gn_table + 32 * lo, hi - lo, self.gn_maxc, self.stream)

  def timed(self, kind, flops, fn, *args):
    """Launch `fn(*args)`; with a profiler attached, bracket it with HIP events on the launch stream."""
    if self.prof is None:
      fn(*args)
    else:
      self.prof.launch(kind, flops, fn, args)

  def v(self, t):
    return None if t is None else self.base[t.space] + 4 * t.off

  def g(self, t):
    if t is None or not t.needs_grad or t.goff is None:
      return None
    if t.space == 'param' and not self.param_grads:
      return None
    return self.gbase[t.space] + 4 * t.goff

  def planes(self, t):
    return self.pl + t.pl_off

  def rec(self, t):
    rt, off = t.pl_rec
    return self.v(rt) + 4 * off

  def make_planes(self, t):
    """fp32 tensor -> planes with a MEASURED scale: one |x| pass into the record, one split pass."""
    n = t.shape[0]
    c = t.shape[1]
    hw = t.numel // (n * c)
    self.lib.amax_partial_f32(self.v(t), t.numel, self.rec(t), self.stream)
    self.lib.split_planes_f32(self.v(t), n, c, hw, self.rec(t), 256, self.planes(t), self.stream)


class Op:
  """Base class.  Subclasses list `inputs` (Tensors whose gradients they may write)."""
  inputs = ()

  def plan_backward(self):
    """Called in backward order: fix beta (0 = overwrite, 1 = accumulate) per gradient target."""
    self.beta = {}
    for t in self.inputs:
      if t is None or not t.needs_grad or t.space != 'act':
        continue
      if id(t) in self.beta:
        continue
      self.beta[id(t)] = 1.0 if (t.seen > 0 or t.external_grad) else 0.0
      t.seen += 1

  def b(self, t):
    return self.beta.get(id(t), 1.0)

  def forward(self, rt):
    raise NotImplementedError

  def backward(self, rt):
    raise NotImplementedError

  def ws_bytes(self, lib):
    return 0

  def bwd_launches(self):
    """False when this op's backward is known at plan time to launch nothing (the executor does not capture / replay
    a backward segment made of such ops only)."""
    xs = [t for t in self.inputs if t is not None]
    return not xs or any(t.needs_grad for t in xs) or any(
      isinstance(v, Tensor) and v.space == 'param' and v.needs_grad for v in vars(self).values())


# ------------------------------------------------------------------------------------------------
# ops
# ------------------------------------------------------------------------------------------------
class GroupNormAct(Op):
  """y = dropout(act(GroupNorm(cat(x1, x2))))  -- nn.GroupNorm + SiLU + Dropout call sites,
  reference models/layerspp.py:256,277-278 (ResnetBlockBigGANpp), :90 (AttnBlockpp, act=False)."""

  def __init__(self, g, x1, x2, gamma, beta_t, groups, eps, act, drop_p, name):
    self.x1, self.x2, self.gamma, self.beta_t = x1, x2, gamma, beta_t
    N, C1, H, W = x1.shape
    C2 = x2.shape[1] if x2 is not None else 0
    self.N, self.C1, self.C2, self.HW, self.G = N, C1, C2, H * W, groups
    self.eps, self.act, self.drop_p = eps, int(act), float(drop_p)
    self.y = g.new((N, C1 + C2, H, W), name=name)
    self.mean = g.new((N * groups,), needs_grad=False, name=name + '.mean')
    self.rstd = g.new((N * groups,), needs_grad=False, name=name + '.rstd')
    self.inputs = (x1, x2)
    self.op_id = g.next_id()
    self.fused = False        # the library's one-pass GroupNorm -> planes kernel takes this shape (Graph.finalize)
    self.fold_off = None      # float offset of this layer's [N][C][2] partial sums in the program's arena (Graph.finalize)
    self.fold_index = None    # its entry in the fold table (backward order)
    # Graph._plan_dy_producers: the convolution whose output this layer normalises and nobody else reads.  Its output
    # gradient IS this layer's dx1, so the backward kernel leaves the bias / time-embedding sums and the |dy| scale
    # record behind (stk_gn_bwd_out_f32) and the convolution's backward makes no pass of its own over dy.
    self.dy_cons = None
    self.add_from = None      # Graph._plan_res_via: the convolution whose output gradient / out_div this layer adds into dx1
    # Graph._plan_x_records: the block's 1x1 shortcut convolution reads this layer's SOURCE tensors as fp32 operands of the
    # split kernels; the one-pass forward leaves their |x| scale records in that convolution's amax buffer
    self.xmax_for = None

  def _p(self, rt):
    return self.drop_p if rt.training else 0.0

  def forward(self, rt):
    seed = (rt.seed + 0x9E3779B1 * self.op_id) & 0xFFFFFFFFFFFFFFFF
    y = self.y
    if y.pl_maker is self:
      # consumers take this output as planes (include/stk.h "Planes"): normalise, bound record and split in one call
      # (one pass over x for the shapes the library fuses); the scale is the a-priori bound of the affine parameters.
      # The fp32 copy is written only if somebody reads it (the fused shapes accept y = NULL).
      need_f32 = y.f32_fwd or (rt.with_backward and y.f32_bwd) or not self.fused
      cv = self.xmax_for
      if cv is not None:
        rt.lib.gn_fwd_pl_max_f32(rt.v(self.x1), self.C1, rt.v(self.x2), self.C2, rt.v(self.gamma), rt.v(self.beta_t),
                                 rt.v(y) if need_f32 else None, rt.planes(y), rt.rec(y), rt.v(self.mean), rt.v(self.rstd),
                                 self.N, self.HW, self.G, self.eps, self.act, self._p(rt), seed, rt.seed_dev, rt.ws,
                                 rt.v(cv.amax), rt.v(cv.amax) + 4 * 256 if self.x2 is not None else None, rt.stream)
        return
      rt.lib.gn_fwd_pl_f32(rt.v(self.x1), self.C1, rt.v(self.x2), self.C2, rt.v(self.gamma), rt.v(self.beta_t),
                           rt.v(y) if need_f32 else None, rt.planes(y), rt.rec(y), rt.v(self.mean), rt.v(self.rstd), self.N, self.HW, self.G,
                           self.eps, self.act, self._p(rt), seed, rt.seed_dev, rt.ws, rt.stream)
      return
    rt.lib.gn_fwd_f32(rt.v(self.x1), self.C1, rt.v(self.x2), self.C2, rt.v(self.gamma), rt.v(self.beta_t),
                      rt.v(self.y), rt.v(self.mean), rt.v(self.rstd), self.N, self.HW, self.G, self.eps,
                      self.act, self._p(rt), seed, rt.seed_dev, rt.ws, rt.stream)

  def backward(self, rt):
    dgamma, dbeta, ws = rt.g(self.gamma), rt.g(self.beta_t), rt.ws
    cons = self.dy_cons
    if rt.gn_table and self.fold_index is not None and (dgamma is not None or dbeta is not None):
      # leave the per-(sample, channel) sums in this layer's own slot; one launch folds a whole segment's layers
      dgamma = dbeta = None
      ws = rt.gnpart + 4 * self.fold_off
      rt.defer_fold(self.fold_index)
    adder = self.add_from
    if cons is not None or adder is not None:
      dsum = dtemb = damax = add = None
      out_scale = add_scale = 1.0
      tstride = 0
      if cons is not None:
        assert rt.gn_table, 'the by-products of the GroupNorm backward ride on the batched folds'
        out_scale = 1.0 / cons.out_div
        if cons.bsum_index is not None and rt.param_grads:
          dsum = rt.gnpart + 4 * cons.bsum_off
          rt.defer_fold(cons.bsum_index)
        if cons.temb is not None and cons.temb.needs_grad:
          dtemb, tstride = rt.g(cons.temb) + 4 * cons.temb_col, cons.temb_stride
        damax = rt.v(cons.dy_peer.amax if cons.dy_peer is not None else cons.amax) + 4 * 512
      if adder is not None:
        add, add_scale = rt.g(adder.y), 1.0 / adder.out_div
      rt.lib.gn_bwd_out_f32(rt.g(self.y), rt.v(self.x1), self.C1, rt.v(self.x2), self.C2, rt.v(self.gamma), rt.v(self.beta_t),
                            rt.v(self.mean), rt.v(self.rstd), rt.g(self.x1), self.b(self.x1), rt.g(self.x2),
                            self.b(self.x2) if self.x2 is not None else 0.0, dgamma, dbeta, ws, self.N, self.HW, self.G,
                            self.act, self._p(rt), (rt.seed + 0x9E3779B1 * self.op_id) & 0xFFFFFFFFFFFFFFFF, rt.seed_dev,
                            add, add_scale, dsum, out_scale, dtemb, tstride, damax, rt.stream)
      return
    rt.lib.gn_bwd_f32(rt.g(self.y), rt.v(self.x1), self.C1, rt.v(self.x2), self.C2,
                      rt.v(self.gamma), rt.v(self.beta_t), rt.v(self.mean), rt.v(self.rstd),
                      rt.g(self.x1), self.b(self.x1), rt.g(self.x2), self.b(self.x2) if self.x2 is not None else 0.0,
                      dgamma, dbeta, ws,
                      self.N, self.HW, self.G, self.act, self._p(rt),
                      (rt.seed + 0x9E3779B1 * self.op_id) & 0xFFFFFFFFFFFFFFFF, rt.seed_dev, rt.stream)

  def ws_bytes(self, lib):
    return max(int(lib.gn_ws_bytes(self.N, self.C1 + self.C2, self.HW, self.G)), 4 * 2 * self.N * (self.C1 + self.C2))


class Conv(Op):
  """y = (conv(cat(x1,x2), w) + bias + temb[:, col:col+Cout] + res) / out_div.

  Covers nn.Conv2d 3x3/1x1 (models/layers.py:100-124), NIN (models/layers.py:546-555, w_layout 1),
  the time-embedding add and the skip_rescale combine of ResnetBlockBigGANpp
  (models/layerspp.py:273-287), Combine('sum') (:57-72) and the strided conv of
  conv_downsample_2d (models/up_or_down_sampling.py:178)."""

  def __init__(self, g, x1, x2, w, bias, w_layout, Cout, KH, KW, stride, pad, OH, OW,
               temb=None, temb_col=0, res=None, out_div=1.0, name='conv'):
    self.x1, self.x2, self.w, self.bias = x1, x2, w, bias
    N, C1, H, W = x1.shape
    C2 = x2.shape[1] if x2 is not None else 0
    self.N, self.C1, self.C2, self.H, self.W = N, C1, C2, H, W
    self.Cout, self.KH, self.KW, self.stride, self.pad, self.OH, self.OW = Cout, KH, KW, stride, pad, OH, OW
    self.w_layout = w_layout
    self.temb, self.temb_col = temb, temb_col
    self.temb_stride = temb.shape[1] if temb is not None else 0
    self.res, self.out_div = res, float(out_div)
    self.y = g.new((N, Cout, OH, OW), name=name)
    # partial |x1|, |x2|, |dy| maxima of the split kernels (include/stk.h "amax"): written by this layer's forward /
    # data-gradient calls, reused by its weight gradient; lives with the activations, i.e. per forward context
    # (all layers' buffers are one contiguous block, placed by Graph.finalize: one strided fill zeroes every |dy| third)
    self.amax = Tensor((AMAX,), 'act', None, False, name + '.amax')
    g.conv_amax.append(self.amax)
    self.inputs = (x1, x2, res)
    # algorithmic FLOPs of one launch (1 MAC = 2 FLOP); identical for fwd, dgrad and wgrad
    self.flops = 2.0 * N * OH * OW * Cout * (C1 + C2) * KH * KW

  _VARIANT = {0: 't64', 1: 't128', 3: 't128', 4: 'thin', 5: 'x2'}

  def _kind(self, lib, direction):
    """Kernel label for the profiler: direction, taps and the kernel family csrc/conv.hip picks for this shape
    (t64 / t128 = f32-input MFMA tiles, x3 = bf16 three-way split, x2 = fp16 two-way split)."""
    key = '_kind_' + direction
    k = getattr(self, key, None)
    if k is None:
      v = int(lib.conv2d_variant({'fwd': 0, 'dgrad': 1, 'wgrad': 2}[direction], self.C1, self.C2, self.N, self.H,
                                 self.W, self.Cout, self.OH, self.OW, self.KH, self.KW, self.stride, self.pad,
                                 self.w_layout))
      k = f'conv{self.KH}x{self.KW}.{direction}.{self._VARIANT.get(v, "t64")}'
      setattr(self, key, k)
    return k

  def _label_pl(self, lib, direction):
    """Profiler label of a plane-operand launch: one label per kernel SYMBOL (bench.py's roofline is per kernel):
    fwd / dgrad: '...x2p' = x2d::gemm_kernel<.., EpFwd / EpDgrad>, '...x2p.k' = its K-split form (EpSlab + slab sum, small
    maps), '...x2p.h16 / .h32 / .h64' = x2d::gemm_halo_kernel<W, ..> (the halo-tile GEMM of the 16 / 32 / 64-wide maps); wgrad: '...x2p.w32' / '.w16' / '.w8' / '.w4' = x2w::wgrad_kernel<min(W, 32)>."""
    key = '_label_' + direction
    k = getattr(self, key, None)
    if k is None:
      k = self._kind(lib, direction) + 'p'
      if direction == 'wgrad':
        k += f'.w{min(self.W, 32)}'
      elif hasattr(lib, 'conv2d_pl_ksplit'):
        d = 0 if direction == 'fwd' else 1
        c2 = 0 if d == 0 else self.C2
        if int(lib.conv2d_pl_ksplit(d, self.C1, c2, self.N, self.H, self.W, self.Cout, self.KH, self.KW)) > 1:
          k += '.k'
        elif hasattr(lib, 'conv2d_pl_halo'):
          hw = int(lib.conv2d_pl_halo(d, self.C1, c2, self.N, self.H, self.W, self.Cout, self.KH, self.KW))
          if hw:
            k += f'.h{hw}'
      setattr(self, key, k)
    return k

  def _dims(self):
    return (self.N, self.H, self.W, self.Cout, self.OH, self.OW, self.KH, self.KW, self.stride, self.pad)

  # prepared weights (include/stk.h "Prepared weights"): byte offsets of this layer's blocks in the program's arena,
  # assigned by Graph.finalize; None = this direction of this shape prepares nothing
  wp_off = (None, None)

  def plan_wp(self, lib, offset):
    """Reserve the forward / data-gradient blocks at `offset`; returns the new end of the arena."""
    shape = (self.C1, self.C2, self.N, self.H, self.W, self.Cout, self.KH, self.KW, self.stride, self.pad)
    offs = []
    for direction in (0, 1):
      nb = int(lib.conv2d_wp_bytes(direction, *shape)) if self.OH == self.H and self.OW == self.W else 0
      if nb > 0:
        offs.append(offset)
        offset = _round_up(offset + nb, 256)
      else:
        offs.append(None)
    self.wp_off = tuple(offs)
    return offset

  def _wp(self, rt, direction):
    off = self.wp_off[direction]
    return rt.wp + off if (rt.wp and off is not None) else None

  # shared output gradient (Graph._plan_shared_dy): this layer's residual input `res` is the output of a shortcut
  # convolution nobody else reads, so d(res) = dy / out_div needs no tensor of its own -- `dy_peer` is that convolution
  # (it differentiates from THIS layer's dy with the factor folded into alpha); `dy_from` is the reverse link
  dy_peer = None
  dy_from = None
  # Graph._plan_dy_producers: the GroupNorm that reads this layer's output (and is its only reader) -- its backward leaves
  # this layer's bias / time-embedding sums and |dy| record behind; dyrec = that record (zeroed once per backward),
  # bsum_off / bsum_index = the partial-sum slot and fold-table entry of the bias gradient
  dy_prod = None
  bsum_off = None
  bsum_index = None
  # Graph._plan_res_via: this layer's residual input `res` is the block input x of out = (x + h) / out_div, and x is also
  # what the block's first GroupNorm normalises: that layer's backward adds d(out) / out_div into d(x) on the way
  # (stk_gn_bwd_out_f32 dx1_add), so this layer's backward does not touch d(res)
  res_via = None
  x_from = None        # Graph._plan_x_records: the GroupNorm whose forward leaves this layer's |x1| / |x2| records behind
  # Graph._plan_shared_dy: a 1x1 shortcut convolution differentiates from its peer's output gradient -- and that gradient
  # already exists as planes (the peer's 3x3 data / weight gradient made them): its data gradient reads THOSE through the LDS-
  # DMA kernel instead of splitting the fp32 tensor again in its loader.  'own' = the peer keeps its dy planes (two streams),
  # 'scratch' = the context's shared scratch, still holding them when this layer's backward runs right after the peer's
  peer_planes = None

  # planes (include/stk.h "Planes"): decided by Graph.finalize
  dypl_off = None      # byte offset of this layer's own dy planes in the planes arena (side-stream weight gradients)
  pl_fwd = False       # the forward call reads x1 as planes
  pl_dgrad = False     # the data-gradient call reads dy as planes (made here, into the context's scratch)
  pl_wgrad = False     # the weight-gradient call reads x1 AND dy as planes (3x3 layers whose forward does)
  x_rec_own = False    # x1's scale record is this layer's amax[0:256] (so the weight gradient may reuse it)

  def plan_planes(self, g, lib):
    same = self.stride == 1 and self.OH == self.H and self.OW == self.W and self.pad == self.KH // 2
    if not same:
      return
    dims = (self.N, self.H, self.W, self.Cout, self.KH, self.KW, 1, self.pad)
    t = self.x1
    by_gn = isinstance(t.producer, GroupNormAct)
    # 3x3: the fp32 loader re-splits every element nine times; 1x1: only worth a split pass when GroupNorm makes
    # the planes anyway (attention's q / k / v projections share one input)
    if self.x2 is None and (self.KH == 3 or by_gn) and int(lib.conv2d_pl_ok(0, self.C1, 0, *dims)):
      if t.pl_off is None:
        t.pl_off = g.pl_bytes
        g.pl_bytes += _round_up(int(lib.planes_bytes(self.N, self.C1, self.H * self.W)), 256)
        t.pl_rec = (self.amax, 0)
        t.pl_maker = t.producer if by_gn else self
        self.x_rec_own = True
      self.pl_fwd = True
    needs_dx = (self.x1.needs_grad and self.x1.space == 'act') or (self.x2 is not None and self.x2.needs_grad)
    if needs_dx and self.KH == 3 and int(lib.conv2d_pl_ok(1, self.C1, self.C2, *dims)):
      self.pl_dgrad = True
    if (self.pl_fwd and self.KH == 3 and self.w_layout == 0 and self.w.needs_grad and
        os.environ.get('STK_PLANES_WGRAD', '1') != '0' and hasattr(lib, 'conv2d_wgrad_pl_ok') and
        int(lib.conv2d_wgrad_pl_ok(self.N, self.H, self.W, self.C1, self.Cout))):
      self.pl_wgrad = True
    if self.pl_dgrad or self.pl_wgrad:
      nb = _round_up(int(lib.planes_bytes(self.N, self.Cout, self.OH * self.OW)), 256)
      g.dypl_bytes = max(g.dypl_bytes, nb)
      if self.pl_wgrad and g.own_dypl:
        self.dypl_off = g.pl_bytes
        g.pl_bytes += nb

  def plan_backward(self):
    if self.dy_peer is not None or self.res_via is not None:      # d(res) is not written here: do not count this op as a writer
      saved, self.inputs = self.inputs, (self.x1, self.x2)
      Op.plan_backward(self)
      self.inputs = saved
    else:
      Op.plan_backward(self)

  def forward(self, rt):
    temb = rt.v(self.temb) + 4 * self.temb_col if self.temb is not None else None
    if self.pl_fwd:
      t = self.x1
      if t.pl_maker is self:
        rt.make_planes(t)
      rt.timed(self._label_pl(rt.lib, 'fwd'), self.flops, rt.lib.conv2d_fwd_pl_f32,
               rt.planes(t), rt.rec(t), self.C1, rt.v(self.w), self.w_layout, rt.v(self.bias), temb, self.temb_stride,
               rt.v(self.res), self.out_div, rt.v(self.y), self.N, self.H, self.W, self.Cout, self.KH, self.KW,
               self._wp(rt, 0), rt.ws, rt.ws_bytes, rt.stream)
      return
    rt.timed(self._kind(rt.lib, 'fwd'), self.flops,
             rt.lib.conv2d_fwd_rec_f32 if self.x_from is not None else rt.lib.conv2d_fwd_wp_f32,
             rt.v(self.x1), self.C1, rt.v(self.x2), self.C2, rt.v(self.w), self.w_layout,
             rt.v(self.bias), temb, self.temb_stride, rt.v(self.res), self.out_div,
             rt.v(self.y), *self._dims(), self._wp(rt, 0), rt.v(self.amax), rt.ws, rt.ws_bytes, rt.stream)

  def backward(self, rt):
    gy = rt.g(self.y)
    alpha = 1.0 / self.out_div
    lib = rt.lib
    src = self.dy_from                          # a later layer whose output gradient, times 1 / its out_div, is ours
    if src is not None:
      gy = rt.g(src.y)
      alpha = alpha / src.out_div
    dtemb = None
    if self.temb is not None and self.temb.needs_grad:
      dtemb = rt.g(self.temb) + 4 * self.temb_col
    gb = rt.g(self.bias)
    g1, g2 = rt.g(self.x1), rt.g(self.x2)
    gw = rt.g(self.w)
    pl_dgrad = self.pl_dgrad and (g1 is not None or g2 is not None)
    pl_wgrad = self.pl_wgrad and gw is not None
    dy_rec = rt.v(self.amax) + 4 * 512           # |dy| scale record: planes of dy, reused by the weight gradient
    rec_done = False
    fuse_rec = (pl_dgrad or pl_wgrad) and self.Cout <= 256 and (dtemb is not None or gb is not None)
    res_grad = self.res is not None and self.res.needs_grad
    peer = self.dy_peer
    if self.res_via is not None:
      res_grad = False                           # the block's first GroupNorm adds d(out) / out_div into d(x) itself
    if self.dy_prod is not None:
      # the GroupNorm backward that wrote dy last left the sums (bias: batched fold; time embedding: written) and the
      # record behind -- for the shortcut peer too, whose buffer then holds the one record both layers read
      gb = dtemb = None
      fuse_rec = False
      rec_done = True
      if peer is not None:
        dy_rec = rt.v(peer.amax) + 4 * 512
        res_grad = False
    elif src is not None:
      # the peer's pass over dy already left our bias gradient and our |dy| record behind
      gb = dtemb = None
      fuse_rec = False
      rec_done = True
    elif peer is not None:
      # one pass over dy for both layers: sums -> both bias gradients, maxima -> both records; d(res) is never formed
      lib.bias_grad_amax_dual_f32(gy, self.N, self.Cout, self.OH * self.OW, alpha, dtemb, self.temb_stride, gb, dy_rec,
                                  rt.g(peer.bias), rt.v(peer.amax) + 4 * 512, rt.ws, rt.stream)
      rec_done = True
      res_grad = fuse_rec = False
    if res_grad and fuse_rec and hasattr(lib, 'bias_grad_amax_res_f32'):
      # one pass over dy: bias / time-embedding sums, the |dy| scale record AND the residual branch's gradient
      lib.bias_grad_amax_res_f32(gy, self.N, self.Cout, self.OH * self.OW, alpha, dtemb, self.temb_stride, gb, dy_rec,
                                 rt.g(self.res), self.b(self.res), rt.ws, rt.stream)
      rec_done = True
      res_grad = fuse_rec = False
    if res_grad:
      gr = rt.g(self.res)
      lib.axpby_f32(gy, alpha, gr, self.b(self.res), gr, self.y.numel, rt.stream)
    if fuse_rec:
      # the bias gradient reads all of dy: it leaves the per-channel |dy| maxima behind as the scale record
      lib.bias_grad_amax_f32(gy, self.N, self.Cout, self.OH * self.OW, alpha, dtemb, self.temb_stride, gb, dy_rec,
                             rt.ws, rt.stream)
      rec_done = True
    elif not rec_done and (dtemb is not None or gb is not None):
      lib.bias_grad_f32(gy, self.N, self.Cout, self.OH * self.OW, alpha, dtemb, self.temb_stride, gb,
                        rt.ws, rt.stream)
    # data gradient first: its |dy| maxima are reused by the weight gradient (the two are independent otherwise)
    have = 1 if self._kind(lib, 'fwd').endswith('.x2') else 0
    if self.pl_fwd and not self.x_rec_own:
      have = 0                                   # x1's record lives with another layer: the weight gradient measures
    if src is not None or self.dy_prod is not None:
      have |= 2                                  # the peer / the GroupNorm backward wrote our |dy| record
    if pl_dgrad or pl_wgrad:
      rec = dy_rec
      if not rec_done:
        lib.amax_partial_f32(gy, self.y.numel, rec, rt.stream)
      dypl = rt.pl + self.dypl_off if self.dypl_off is not None else rt.dypl
      lib.split_planes_f32(gy, self.N, self.Cout, self.OH * self.OW, rec, 256, dypl, rt.stream)
      have |= 2
    if src is not None and self.peer_planes is not None and (g1 is not None or g2 is not None):
      # the peer's dy planes (and the record they were scaled with) serve this 1x1 data gradient too
      ppl = rt.pl + src.dypl_off if self.peer_planes == 'own' else rt.dypl
      rt.timed(self._label_pl(lib, 'dgrad'), self.flops, lib.conv2d_dgrad_pl_f32,
               ppl, rt.v(self.amax) + 4 * 512, rt.v(self.w), self.w_layout, g1, self.C1, self.b(self.x1),
               g2, self.C2, self.b(self.x2) if self.x2 is not None else 0.0,
               alpha, self.N, self.H, self.W, self.Cout, self.KH, self.KW, self._wp(rt, 1), rt.ws, rt.ws_bytes, rt.stream)
    elif pl_dgrad:
      rt.timed(self._label_pl(lib, 'dgrad'), self.flops, lib.conv2d_dgrad_pl_f32,
               dypl, rec, rt.v(self.w), self.w_layout, g1, self.C1, self.b(self.x1),
               g2, self.C2, self.b(self.x2) if self.x2 is not None else 0.0,
               alpha, self.N, self.H, self.W, self.Cout, self.KH, self.KW, self._wp(rt, 1), rt.ws, rt.ws_bytes, rt.stream)
      have |= 2
    elif g1 is not None or g2 is not None:
      rt.timed(self._kind(lib, 'dgrad'), self.flops,
               lib.conv2d_dgrad_rec_f32 if (src is not None or self.dy_prod is not None) else lib.conv2d_dgrad_wp_f32,
               gy, rt.v(self.w), self.w_layout, g1, self.C1, self.b(self.x1),
               g2, self.C2, self.b(self.x2) if self.x2 is not None else 0.0,
               alpha, *self._dims(), self._wp(rt, 1), rt.v(self.amax), rt.ws, rt.ws_bytes, rt.stream)
      if self._kind(lib, 'dgrad').endswith('.x2'):
        have |= 2
    rt._cur = self.y.name
    if pl_wgrad and rt.side is not None and rt.prof is None and self.dypl_off is not None:
      rt.side_launch(lib.conv2d_wgrad_pl_f32, rt.planes(self.x1), rt.rec(self.x1), dypl, dy_rec, gw, alpha, rt.ws2,
                     rt.ws_bytes, self.N, self.H, self.W, self.C1, self.Cout)
    elif pl_wgrad:
      # (on the main stream: STK_X2W_WGS_ALONE may ask for more workgroups than the side-stream launch above uses; see _X2W_WGS_ALONE)
      rt.timed(self._label_pl(lib, 'wgrad'), self.flops, lib.conv2d_wgrad_pl_wgs_f32,
               rt.planes(self.x1), rt.rec(self.x1), dypl, dy_rec, gw, alpha, rt.ws, rt.ws_bytes,
               self.N, self.H, self.W, self.C1, self.Cout, _X2W_WGS_ALONE, rt.stream)
    elif gw is not None and rt.side is not None and rt.prof is None:
      # a weight gradient is a leaf of the backward: x, dy and this layer's own records in, dw out
      rt.side_launch(lib.conv2d_wgrad_amax_f32, rt.v(self.x1), self.C1, rt.v(self.x2), self.C2, gy, gw, self.w_layout, alpha,
                     rt.ws2, rt.ws_bytes, *self._dims(), rt.v(self.amax), have)
    elif gw is not None:
      rt.timed(self._kind(lib, 'wgrad'), self.flops, lib.conv2d_wgrad_amax_f32,
               rt.v(self.x1), self.C1, rt.v(self.x2), self.C2, gy, gw, self.w_layout, alpha,
               rt.ws, rt.ws_bytes, *self._dims(), rt.v(self.amax), have, rt.stream)

  def ws_bytes(self, lib):
    shape = (self.C1, self.C2, self.N, self.H, self.W, self.Cout, self.KH, self.KW, self.stride, self.pad)
    pw = int(lib.conv2d_wgrad_pl_ws_bytes(self.N, self.H, self.W, self.C1, self.Cout)) \
        if (self.C2 == 0 and self.KH == 3 and hasattr(lib, 'conv2d_wgrad_pl_ws_bytes')) else 0
    return max(pw, int(lib.conv2d_wgrad_ws_bytes(self.C1, self.C2, self.N, self.Cout, self.OH, self.OW,
                                                 self.KH, self.KW)),
               int(lib.conv2d_fwd_ws_bytes(*shape)), int(lib.conv2d_dgrad_ws_bytes(*shape)),
               4 * self.N * self.Cout)


class Linear(Op):
  """y[B,out] = x[B,in] @ W[out,in]^T + b  -- nn.Linear (models/ncsnpp.py:97-102, layerspp.py:240)."""

  KSLICE = 256       # data gradient of a wide layer: K (= fout) is cut into slices of this many columns ...
  KSPLIT_MIN = 2048  # ... when the layer is at least this wide

  def __init__(self, g, x, w, bias, name='linear'):
    self.x, self.w, self.bias = x, w, bias
    self.B, self.fin = x.shape
    self.fout = w.shape[0]
    self.y = g.new((self.B, self.fout), name=name)
    self.inputs = (x,)
    # dX = dY W has M x N = B x fin outputs (a handful of tiles) and K = fout: for the stacked time-embedding projections
    # (46 Dense_0 layers, fout = 9984) one workgroup per tile would walk all of K alone.  Split K into slices computed as
    # one batched GEMM into slabs, summed in slice order by a second GEMM with a row of ones (deterministic).
    # (slices of 256 columns, or the largest of 128 / 64 / 32 that divides the width: the 256 x 256 net stacks 10880 = 85 x 128)
    self.kslice = next((k for k in (self.KSLICE, 128, 64, 32) if self.fout % k == 0), 0)
    self.ksplit = self.fout // self.kslice if (self.fout >= self.KSPLIT_MIN and self.kslice and x.needs_grad) else 0
    self.ones = g.const(np.ones(self.ksplit, dtype=np.float32)) if self.ksplit else None

  def forward(self, rt):
    B, fin, fout = self.B, self.fin, self.fout
    rt.lib.gemm_f32(rt.v(self.x), fin, 1, 0, rt.v(self.w), 1, fin, 0, rt.v(self.y), fout, 1, 0,
                    rt.v(self.bias), 2 if self.bias is not None else 0, B, fout, fin, 1, 1.0, 0.0, rt.stream)

  def backward(self, rt):
    B, fin, fout = self.B, self.fin, self.fout
    gy = rt.g(self.y)
    lib = rt.lib
    gx = rt.g(self.x)
    if gx is not None and self.ksplit:
      S, ks = self.ksplit, self.kslice
      lib.gemm_f32(gy, fout, 1, ks, rt.v(self.w), fin, 1, ks * fin, rt.ws, fin, 1, B * fin, None, 0,
                   B, fin, ks, S, 1.0, 0.0, rt.stream)
      lib.gemm_f32(rt.v(self.ones), S, 1, 0, rt.ws, B * fin, 1, 0, gx, B * fin, 1, 0, None, 0,
                   1, B * fin, S, 1, 1.0, self.b(self.x), rt.stream)
    elif gx is not None:   # dX[b][k] = sum_n dY[b][n] W[n][k]
      lib.gemm_f32(gy, fout, 1, 0, rt.v(self.w), fin, 1, 0, gx, fin, 1, 0, None, 0,
                   B, fin, fout, 1, 1.0, self.b(self.x), rt.stream)
    gw = rt.g(self.w)
    if gw is not None:   # dW[n][k] += sum_b dY[b][n] X[b][k]
      lib.gemm_f32(gy, 1, fout, 0, rt.v(self.x), fin, 1, 0, gw, fin, 1, 0, None, 0,
                   fout, fin, B, 1, 1.0, 1.0, rt.stream)
    gb = rt.g(self.bias)
    if gb is not None:
      lib.bias_grad_f32(gy, B, fout, 1, 1.0, None, 0, gb, rt.ws, rt.stream)

  def ws_bytes(self, lib):
    return max(4 * self.B * self.fout, 4 * self.ksplit * self.B * self.fin)


class SiLU(Op):
  """The model's activation (layers.get_act, models/layers.py:29-41) on a plain tensor; `code` = STK_ACT_* of include/stk.h
  (1 = swish / nn.SiLU, what every shipped config uses; 2 ReLU, 3 LeakyReLU(0.2), 4 ELU)."""

  def __init__(self, g, x, name='silu', code=1):
    self.x = x
    self.code = int(code)
    self.y = g.new(x.shape, name=name)
    self.inputs = (x,)

  def forward(self, rt):
    if self.code == 1:
      rt.lib.silu_fwd_f32(rt.v(self.x), rt.v(self.y), self.x.numel, rt.stream)
    else:
      rt.lib.act_fwd_f32(rt.v(self.x), rt.v(self.y), self.x.numel, self.code, rt.stream)

  def backward(self, rt):
    gx = rt.g(self.x)
    if gx is None:
      return
    if self.code == 1:
      rt.lib.silu_bwd_f32(rt.v(self.x), rt.g(self.y), gx, self.b(self.x), self.x.numel, rt.stream)
    else:
      rt.lib.act_bwd_f32(rt.v(self.x), rt.g(self.y), gx, self.b(self.x), self.x.numel, self.code, rt.stream)


class TimestepEmbedding(Op):
  """layers.get_timestep_embedding (models/layers.py:515-529); no parameters, no gradient."""

  def __init__(self, g, t, dim, max_positions=10000, name='temb.pos'):
    import torch
    self.t, self.dim = t, dim
    half = dim // 2
    # frequency table evaluated exactly as the reference does on the host (torch.exp of an f32 ramp)
    e = math.log(max_positions) / (half - 1)
    self.freqs = g.const(torch.exp(torch.arange(half, dtype=torch.float32) * -e).numpy())
    self.y = g.new((t.shape[0], dim), needs_grad=False, name=name)

  def forward(self, rt):
    rt.lib.timestep_embedding_f32(rt.v(self.t), rt.v(self.freqs), rt.v(self.y), self.t.shape[0], self.dim, rt.stream)

  def backward(self, rt):
    pass

  def bwd_launches(self):
    return False


class FourierEmbedding(Op):
  """GaussianFourierProjection (models/layerspp.py:45-54); W is frozen (requires_grad=False)."""

  def __init__(self, g, x, W, name='temb.fourier'):
    self.x, self.W = x, W
    self.nf = W.shape[0]
    self.y = g.new((x.shape[0], 2 * self.nf), needs_grad=False, name=name)

  def forward(self, rt):
    rt.lib.fourier_embedding_f32(rt.v(self.x), rt.v(self.W), rt.v(self.y), self.x.shape[0], self.nf, rt.stream)

  def backward(self, rt):
    pass

  def bwd_launches(self):
    return False


class Affine(Op):
  """y = a*x + b (the `2x - 1` recentring, models/ncsnpp.py:296-298)."""

  def __init__(self, g, x, a, b, name='affine'):
    self.x, self.a, self.bb = x, float(a), float(b)
    self.y = g.new(x.shape, needs_grad=x.needs_grad, name=name)
    self.inputs = (x,)

  def forward(self, rt):
    rt.lib.affine_f32(rt.v(self.x), self.a, self.bb, rt.v(self.y), self.x.numel, rt.stream)

  def backward(self, rt):
    gx = rt.g(self.x)
    if gx is not None:
      rt.lib.axpby_f32(rt.g(self.y), self.a, gx, self.b(self.x), gx, self.x.numel, rt.stream)


class Concat(Op):
  """torch.cat([a, b], dim=1) as a tensor of its own (Combine(method='cat'), models/layerspp.py:57-72).  The skip connections of the
  up path never need this -- GroupNorm and the convolutions take two sources -- but a concatenation that is itself pushed on the
  skip stack and concatenated again does."""

  def __init__(self, g, a, b, name='cat'):
    self.a, self.bt = a, b
    N, Ca, H, W = a.shape
    Cb = b.shape[1]
    self.N, self.Ca, self.Cb, self.HW = N, Ca, Cb, H * W
    self.y = g.new((N, Ca + Cb, H, W), name=name)
    self.inputs = (a, b)

  def forward(self, rt):
    rt.lib.concat_f32(rt.v(self.a), self.Ca, rt.v(self.bt), self.Cb, rt.v(self.y), self.N, self.HW, rt.stream)

  def backward(self, rt):
    ga, gb = rt.g(self.a), rt.g(self.bt)
    if ga is None and gb is None:
      return
    rt.lib.concat_bwd_f32(rt.g(self.y), ga, self.b(self.a) if ga is not None else 0.0, self.Ca,
                          gb, self.b(self.bt) if gb is not None else 0.0, self.Cb, self.N, self.HW, rt.stream)


class FixedFourier(Op):
  """layerspp.FixedFouriereProjection (models/layerspp.py:31-43): y = cat(x, sin / cos of 128 pi x and 256 pi x) -> 5 C channels."""

  def __init__(self, g, x, name='fixed_fourier'):
    self.x = x
    N, C, H, W = x.shape
    self.N, self.C, self.HW = N, C, H * W
    self.y = g.new((N, 5 * C, H, W), needs_grad=x.needs_grad, name=name)
    self.inputs = (x,)

  def forward(self, rt):
    rt.lib.fixed_fourier_fwd_f32(rt.v(self.x), rt.v(self.y), self.N, self.C, self.HW, rt.stream)

  def backward(self, rt):
    gx = rt.g(self.x)
    if gx is not None:
      rt.lib.fixed_fourier_bwd_f32(rt.v(self.x), rt.g(self.y), gx, self.b(self.x), self.N, self.C, self.HW, rt.stream)


class ResampleNaive(Op):
  """naive_upsample_2d (2x2 repeat) / naive_downsample_2d (2x2 mean),
  models/up_or_down_sampling.py:59-69."""

  def __init__(self, g, x, up, name='resample'):
    self.x, self.up = x, bool(up)
    N, C, H, W = x.shape
    self.planes, self.H, self.W = N * C, H, W
    oshape = (N, C, H * 2, W * 2) if up else (N, C, H // 2, W // 2)
    self.y = g.new(oshape, needs_grad=x.needs_grad, name=name)
    self.inputs = (x,)

  def forward(self, rt):
    rt.lib.resample_naive_f32(rt.v(self.x), rt.v(self.y), self.planes, self.H, self.W,
                              0 if self.up else 1, 1.0, 0.0, rt.stream)

  def backward(self, rt):
    gx = rt.g(self.x)
    if gx is None:
      return
    if self.up:   # d/dx of repeat = 2x2 sum = 4 * mean
      rt.lib.resample_naive_f32(rt.g(self.y), gx, self.planes, 2 * self.H, 2 * self.W, 1, 4.0,
                                self.b(self.x), rt.stream)
    else:         # d/dx of mean = 0.25 * repeat
      rt.lib.resample_naive_f32(rt.g(self.y), gx, self.planes, self.H // 2, self.W // 2, 0, 0.25,
                                self.b(self.x), rt.stream)


class UpFirDn(Op):
  """op.upfirdn2d on [N,C,H,W] (op/upfirdn2d.py:88-156).  Backward = the same operator with the
  flipped taps, up<->down swapped and g_pad (op/upfirdn2d.py:111-114)."""

  def __init__(self, g, x, taps, up, down, pad, name='upfirdn'):
    self.x = x
    taps = np.asarray(taps, dtype=np.float32)
    self.kh, self.kw = taps.shape
    self.k = g.const(taps)
    self.kflip = g.const(taps[::-1, ::-1].copy())
    self.up, self.down, self.pad0, self.pad1 = int(up), int(down), int(pad[0]), int(pad[1])
    N, C, H, W = x.shape
    self.major, self.H, self.W = N * C, H, W
    self.OH = (H * up + pad[0] + pad[1] - self.kh) // down + 1
    self.OW = (W * up + pad[0] + pad[1] - self.kw) // down + 1
    self.gpad0_x = self.kw - self.pad0 - 1
    self.gpad0_y = self.kh - self.pad0 - 1
    self.gpad1_x = W * up - self.OW * down + self.pad0 - up + 1
    self.gpad1_y = H * up - self.OH * down + self.pad0 - up + 1
    self.y = g.new((N, C, self.OH, self.OW), needs_grad=x.needs_grad, name=name)
    self.inputs = (x,)

  def forward(self, rt):
    rt.lib.upfirdn2d_f32(rt.v(self.x), rt.v(self.k), rt.v(self.y), self.major, self.H, self.W, 1,
                         self.kh, self.kw, self.up, self.up, self.down, self.down,
                         self.pad0, self.pad1, self.pad0, self.pad1, rt.stream)

  def backward(self, rt):
    gx = rt.g(self.x)
    if gx is None:
      return
    rt.lib.upfirdn2d_acc_f32(rt.g(self.y), rt.v(self.kflip), gx, self.b(self.x), self.major, self.OH, self.OW, 1,
                             self.kh, self.kw, self.down, self.down, self.up, self.up,
                             self.gpad0_x, self.gpad1_x, self.gpad0_y, self.gpad1_y, rt.stream)


class AddDiv(Op):
  """out = (a + b) / div -- skip_rescale combines (models/ncsnpp.py:340-343,400-403) and the
  plain pyramid sum (models/ncsnpp.py:396) with div = 1."""

  def __init__(self, g, a, b, div, name='add'):
    self.a, self.bt, self.div = a, b, float(div)
    self.y = g.new(a.shape, name=name)
    self.inputs = (a, b)

  def forward(self, rt):
    rt.lib.add_div_f32(rt.v(self.a), rt.v(self.bt), self.div, rt.v(self.y), self.a.numel, rt.stream)

  def backward(self, rt):
    gy = rt.g(self.y)
    for t in (self.a, self.bt):
      gt = rt.g(t)
      if gt is not None:
        rt.lib.axpby_f32(gy, 1.0 / self.div, gt, self.b(t), gt, t.numel, rt.stream)
        # a tensor added to itself would need beta=1 on the second pass
        if self.a is self.bt:
          rt.lib.axpby_f32(gy, 1.0 / self.div, gt, 1.0, gt, t.numel, rt.stream)
          break


class AttentionCore(Op):
  """o = softmax(q^T k / sqrt(C)) applied to v -- the two einsums and the softmax of AttnBlockpp
  (models/layerspp.py:95-99), single head of dimension C.  q, k, v are either three [B,C,H,W] tensors or, with
  `qkv`, the channel slices [0,C), [C,2C), [2C,3C) of ONE [B,3C,H,W] tensor (the stacked projection of
  AttnBlockpp.emit), read and differentiated in place through a batch stride of 3 C T.

  Shapes the library's fused kernels take (stk_attention_ok: every shipped config) run as one launch per direction pair
  with no [B,T,T] matrix in memory; the rest as batched GEMMs around stk_softmax_*.  STK_ATTN_FUSED=0 forces the latter."""

  def __init__(self, g, q, k, v, name='attn', qkv=None):
    self.qkv = qkv
    if qkv is not None:
      B, C3, H, W = qkv.shape
      C = C3 // 3
      self.q = self.k = self.vv = None
      self.inputs = (qkv,)
    else:
      self.q, self.k, self.vv = q, k, v
      B, C, H, W = q.shape
      self.inputs = (q, k, v)
    self.B, self.C, self.T = B, C, H * W
    self.bs = (3 if qkv is not None else 1) * C * self.T       # floats between consecutive images of q / k / v
    self.scale = float(int(C) ** (-0.5))
    lib = g.lib
    self.fused = bool(lib is not None and os.environ.get('STK_ATTN_FUSED', '1') != '0' and hasattr(lib, 'attention_ok') and
                      int(lib.attention_ok(B, C, self.T)))
    if self.fused:
      self.lse = g.new((B, self.T), needs_grad=False, name=name + '.lse')
      self.delta = g.new((B, self.T), needs_grad=False, name=name + '.delta')
      self.rec = g.new((1024,), needs_grad=False, name=name + '.rec')      # scale records of q, k, v, do
    else:
      self.s = g.new((B, self.T, self.T), needs_grad=False, name=name + '.s')
      self.p = g.new((B, self.T, self.T), needs_grad=False, name=name + '.p')
    self.o = g.new((B, C, H, W), name=name + '.o')
    self.y = self.o
    # algorithmic FLOPs: forward 2 GEMMs, backward 4 (+ 3 recomputed by the fused kernels, not counted)
    self.flops = 2.0 * 2.0 * B * self.T * self.T * C

  def _qkv(self, rt):
    """addresses of q, k, v and of their gradients (None = not wanted) + the gradients' beta"""
    n4 = 4 * self.C * self.T
    if self.qkv is not None:
      base, gbase = rt.v(self.qkv), rt.g(self.qkv)
      grads = (None, None, None) if gbase is None else (gbase, gbase + n4, gbase + 2 * n4)
      beta = self.b(self.qkv)
      return (base, base + n4, base + 2 * n4), grads, (beta, beta, beta)
    return ((rt.v(self.q), rt.v(self.k), rt.v(self.vv)), (rt.g(self.q), rt.g(self.k), rt.g(self.vv)),
            (self.b(self.q), self.b(self.k), self.b(self.vv)))

  def forward(self, rt):
    B, C, T, bs = self.B, self.C, self.T, self.bs
    lib = rt.lib
    (q, k, v), _, _ = self._qkv(rt)
    if self.fused:
      rt.timed('attention.fwd.x2', self.flops, lib.attention_fwd_f32, q, k, v, bs, rt.v(self.o),
               rt.v(self.lse), rt.v(self.rec), B, C, T, self.scale, rt.stream)
      return
    # S[b][t][t'] = sum_c Q[b][c][t] K[b][c][t']
    lib.gemm_f32(q, 1, T, bs, k, T, 1, bs, rt.v(self.s), T, 1, T * T,
                 None, 0, T, T, C, B, 1.0, 0.0, rt.stream)
    lib.softmax_fwd_f32(rt.v(self.s), rt.v(self.p), B * T, T, self.scale, rt.stream)
    # O[b][c][t] = sum_t' V[b][c][t'] P[b][t][t']
    lib.gemm_f32(v, T, 1, bs, rt.v(self.p), 1, T, T * T, rt.v(self.o), T, 1, C * T,
                 None, 0, C, T, T, B, 1.0, 0.0, rt.stream)

  def backward(self, rt):
    B, C, T, bs = self.B, self.C, self.T, self.bs
    lib = rt.lib
    go = rt.g(self.o)
    (q, k, v), (gq, gk, gv), (bq, bk, bv) = self._qkv(rt)
    if gq is None and gk is None and gv is None:
      return
    if self.fused:
      # the kernels write all three gradients; one that nobody asked for goes to the workspace (separate tensors only)
      n4 = 4 * B * C * T
      gs = bs if self.qkv is not None else C * T
      spare = [rt.ws + i * n4 for i in range(3)]
      rt.timed('attention.bwd.x2', 2.0 * self.flops, lib.attention_bwd_f32, q, k, v, bs, go,
               rt.v(self.lse), rt.v(self.rec), rt.v(self.delta),
               gq if gq is not None else spare[0], bq if gq is not None else 0.0,
               gk if gk is not None else spare[1], bk if gk is not None else 0.0,
               gv if gv is not None else spare[2], bv if gv is not None else 0.0,
               gs, B, C, T, self.scale, rt.stream)
      return
    gs = bs if self.qkv is not None else C * T
    dp = rt.v(self.s)   # S is dead after the forward softmax: reuse it for dP, then dS
    # dP[b][t][t'] = sum_c dO[b][c][t] V[b][c][t']
    lib.gemm_f32(go, 1, T, C * T, v, T, 1, bs, dp, T, 1, T * T,
                 None, 0, T, T, C, B, 1.0, 0.0, rt.stream)
    if gv is not None:  # dV[b][c][t'] = sum_t dO[b][c][t] P[b][t][t']
      lib.gemm_f32(go, T, 1, C * T, rt.v(self.p), T, 1, T * T, gv, T, 1, gs,
                   None, 0, C, T, T, B, 1.0, bv, rt.stream)
    lib.softmax_bwd_f32(rt.v(self.p), dp, dp, B * T, T, self.scale, rt.stream)   # dS in place
    if gq is not None:  # dQ[b][c][t] = sum_t' K[b][c][t'] dS[b][t][t']
      lib.gemm_f32(k, T, 1, bs, dp, 1, T, T * T, gq, T, 1, gs,
                   None, 0, C, T, T, B, 1.0, bq, rt.stream)
    if gk is not None:  # dK[b][c][t'] = sum_t Q[b][c][t] dS[b][t][t']
      lib.gemm_f32(q, T, 1, bs, dp, T, 1, T * T, gk, T, 1, gs,
                   None, 0, C, T, T, B, 1.0, bk, rt.stream)

  def ws_bytes(self, lib):
    return 3 * 4 * self.B * self.C * self.T if (self.fused and self.qkv is None) else 0


class RowScale(Op):
  """out[n] = x[n] / s[n]  (scale_by_sigma, models/ncsnpp.py:428-430); s carries no gradient."""

  def __init__(self, g, x, s, name='rowscale'):
    self.x, self.s = x, s
    self.N = x.shape[0]
    self.inner = x.numel // self.N
    self.y = g.new(x.shape, name=name)
    self.inputs = (x,)

  def forward(self, rt):
    rt.lib.rowscale_f32(rt.v(self.x), rt.v(self.s), rt.v(self.y), self.N, self.inner, 1, rt.stream)

  def backward(self, rt):
    gx = rt.g(self.x)
    if gx is None:
      return
    if self.b(self.x) == 0.0:
      rt.lib.rowscale_f32(rt.g(self.y), rt.v(self.s), gx, self.N, self.inner, 1, rt.stream)
    else:
      # accumulate through a scratch row-scale in the workspace
      rt.lib.rowscale_f32(rt.g(self.y), rt.v(self.s), rt.ws, self.N, self.inner, 1, rt.stream)
      rt.lib.axpby_f32(rt.ws, 1.0, gx, 1.0, gx, self.x.numel, rt.stream)

  def ws_bytes(self, lib):
    return 4 * self.x.numel


class ZeroXRecords(Op):
  """First op of the plan: zeroes the |x1| / |x2| thirds of every convolution's amax buffer -- the records the GroupNorm
  forward kernels fill by atomic maximum (Graph._plan_x_records); every other user writes its record after this."""

  def __init__(self, block, count):
    self.block, self.count = block, count
    self.y = block

  def forward(self, rt):
    rt.lib.fill_strided_f32(rt.v(self.block), 0.0, self.count, 512, AMAX, rt.stream)

  def backward(self, rt):
    pass

  def bwd_launches(self):
    return False


class ZeroRecords(Op):
  """Last op of the plan, i.e. first of the backward: zeroes the |dy| third of every convolution's amax buffer -- the scale
  records that the GroupNorm backward kernels fill by atomic maximum (Graph._plan_dy_producers) -- in one launch."""

  def __init__(self, block, count):
    self.block, self.count = block, count
    self.y = block

  def forward(self, rt):
    pass

  def backward(self, rt):
    rt.lib.fill_strided_f32(rt.v(self.block) + 4 * 512, 0.0, self.count, AMAX - 512, AMAX, rt.stream)


# ------------------------------------------------------------------------------------------------
# graph
# ------------------------------------------------------------------------------------------------
class Graph:
  """Builder: allocates symbolic tensors, records ops, then plans gradient buffers."""

  def __init__(self, flat, lib=None):
    self.flat = flat                  # engine.flat.FlatParams (parameter -> flat offset)
    self.lib = lib                    # the backend the plan is made for (ops may ask it which kernels take a shape)
    self.ops = []
    self.tensors = []
    self.conv_amax = []               # the convolutions' 768-float amax buffers (offsets assigned by finalize)
    self.act_size = 0
    self.gact_size = 0
    self.const_chunks = []
    self.const_size = 0
    self.inputs = {}
    self.output = None
    self._id = 0
    self._params = {}

  def next_id(self):
    self._id += 1
    return self._id

  def new(self, shape, needs_grad=True, name=''):
    t = Tensor(shape, 'act', self.act_size, needs_grad, name)
    self.act_size += _round_up(t.numel)
    self.tensors.append(t)
    return t

  def input(self, key, shape, needs_grad=False):
    t = self.new(shape, needs_grad=needs_grad, name='in.' + key)
    self.inputs[key] = t
    return t

  def const(self, array):
    array = np.ascontiguousarray(array, dtype=np.float32)
    t = Tensor(array.shape, 'const', self.const_size, False, 'const')
    self.const_chunks.append((self.const_size, array.reshape(-1)))
    self.const_size += _round_up(array.size)
    return t

  def param(self, p):
    """Tensor bound to an nn.Parameter's slot in the flat parameter / gradient buffers."""
    if p is None:
      return None
    t = self._params.get(id(p))
    if t is None:
      off, trainable = self.flat.offset_of(p)
      t = Tensor(tuple(p.shape), 'param', off, trainable, 'param')
      t.goff = off if trainable else None
      self._params[id(p)] = t
    return t

  def add(self, op):
    self.ops.append(op)
    if op.y.producer is None:
      op.y.producer = op
    return op.y

  # -- convenience emitters -------------------------------------------------------------------
  # STK_ACT_* code of the model's nonlinearity (set by the model before it emits; 1 = swish)
  act_code = 1

  def gn_act(self, x1, x2, gn, act=True, drop_p=0.0, name='gn'):
    op = GroupNormAct(self, x1, x2, self.param(gn.weight), self.param(gn.bias), gn.num_groups, gn.eps,
                      self.act_code if act else 0, drop_p, name)
    return self.add(op)

  def conv(self, x1, x2, weight, bias, w_layout=0, stride=1, pad=None, out_hw=None,
           temb=None, temb_col=0, res=None, out_div=1.0, name='conv'):
    if w_layout == 0:
      Cout, _, KH, KW = weight.shape
    else:
      _, Cout = weight.shape
      KH = KW = 1
    if pad is None:
      pad = KH // 2
    # the kernels take the channel counts from the tensors, the weight's own width is never read: a mismatch (hand-kept
    # channel bookkeeping of the emitters) would convolve with the wrong rows of w silently -- the reference raises here
    cin_w = weight.shape[1] if w_layout == 0 else weight.shape[0]
    cin_x = x1.shape[1] + (x2.shape[1] if x2 is not None else 0)
    if cin_w != cin_x:
      raise RuntimeError(f'{name}: expected input with {cin_w} channels (weight {tuple(weight.shape)}), got {cin_x}')
    H, W = x1.shape[2], x1.shape[3]
    if out_hw is None:
      OH = (H + 2 * pad - KH) // stride + 1
      OW = (W + 2 * pad - KW) // stride + 1
    else:
      OH, OW = out_hw
    op = Conv(self, x1, x2, self.param(weight), self.param(bias), w_layout, Cout, KH, KW, stride, pad, OH, OW,
              temb=temb, temb_col=temb_col, res=res, out_div=out_div, name=name)
    return self.add(op)

  def conv1x1_t(self, x1, w_tensor, b_tensor, Cout, name='conv'):
    """1x1 convolution (NIN layout, w[Cin][Cout]) on graph tensors that stand for SEVERAL stacked parameters."""
    H, W = x1.shape[2], x1.shape[3]
    if w_tensor.shape[0] != x1.shape[1]:
      raise RuntimeError(f'{name}: expected input with {w_tensor.shape[0]} channels, got {x1.shape[1]}')
    return self.add(Conv(self, x1, None, w_tensor, b_tensor, 1, Cout, 1, 1, 1, 0, H, W, name=name))

  def linear(self, x, weight, bias, name='linear'):
    return self.add(Linear(self, x, self.param(weight), self.param(bias), name))

  def linear_t(self, x, w_tensor, b_tensor, name='linear'):
    return self.add(Linear(self, x, w_tensor, b_tensor, name))

  def silu(self, x, name='silu'):
    return self.add(SiLU(self, x, name, self.act_code))

  # -- planning -----------------------------------------------------------------------------------
  def finalize(self, output, lib):
    self.output = output
    output.external_grad = True
    self.amax_block = self.new((AMAX * max(len(self.conv_amax), 1),), needs_grad=False, name='conv.amax')
    for i, t in enumerate(self.conv_amax):
      t.off = self.amax_block.off + AMAX * i
    for t in self.tensors:
      if t.needs_grad:
        t.goff = self.gact_size
        self.gact_size += _round_up(t.numel)
    self.ws_bytes = max([256] + [op.ws_bytes(lib) for op in self.ops])
    self.ws_bytes = _round_up(self.ws_bytes, 256)
    # prepared-weight arena: one block per conv layer and direction that runs on the split kernel
    self.wp_bytes = 0
    for op in self.ops:
      if isinstance(op, Conv):
        self.wp_bytes = op.plan_wp(lib, self.wp_bytes)
    # planes arena: pre-split copies of the activations the split convolutions read (STK_PLANES=0: none, every call
    # takes fp32 operands -- a debugging switch)
    self.pl_bytes = 0
    self.dypl_bytes = 0
    # weight gradients on a side stream (engine/executor.py): every such layer keeps the planes of its output gradient
    self.own_dypl = os.environ.get('STK_WGRAD_STREAM', '1') != '0' and bool(getattr(lib, 'is_device', False))
    if os.environ.get('STK_PLANES', '1') != '0' and hasattr(lib, 'conv2d_pl_ok'):
      for op in self.ops:
        if isinstance(op, Conv):
          op.plan_planes(self, lib)
      self._plan_f32_copies(lib)
      self._plan_x_records(lib)
    self._plan_shared_dy(lib)
    fold_batch = os.environ.get('STK_GN_FOLD_BATCH', '1') != '0' and hasattr(lib, 'gn_param_grad_batch')
    if hasattr(lib, 'gn_bwd_out_f32'):
      self._plan_res_via(lib)
      if fold_batch:
        self._plan_dy_producers(lib)
    # deferred parameter-gradient folds: a slot of [N][C][2] partial sums per GroupNorm layer (and per convolution bias
    # served by a GroupNorm backward), table entries (slot offset, dgamma offset, dbeta offset, N, C) in backward order
    # (STK_GN_FOLD_BATCH=0: every layer folds its own sums -- a debugging switch, results are bit-identical)
    self.gnpart_size = 0
    self.gn_folds = []
    if fold_batch:
      for op in reversed(self.ops):
        if not isinstance(op, GroupNormAct):
          continue
        cons = op.dy_cons
        if op.gamma.needs_grad or op.beta_t.needs_grad or cons is not None:
          op.fold_off = self.gnpart_size
          op.fold_index = len(self.gn_folds)
          self.gnpart_size += _round_up(max(int(lib.gn_ws_bytes(op.N, op.C1 + op.C2, op.HW, op.G)) // 4,
                                            2 * op.N * (op.C1 + op.C2)))
          self.gn_folds.append((op.fold_off, op.gamma.goff if op.gamma.needs_grad else None,
                                op.beta_t.goff if op.beta_t.needs_grad else None, op.N, op.C1 + op.C2))
        if cons is not None:
          # one entry, two targets: component 0 -> the convolution's bias gradient, component 1 -> its shortcut peer's
          own = cons.bias.goff if cons.bias is not None and cons.bias.needs_grad else None
          peer = cons.dy_peer
          other = peer.bias.goff if peer is not None and peer.bias is not None and peer.bias.needs_grad else None
          if own is not None or other is not None:
            cons.bsum_off = self.gnpart_size
            cons.bsum_index = len(self.gn_folds)
            self.gnpart_size += _round_up(2 * cons.N * cons.Cout)
            self.gn_folds.append((cons.bsum_off, other, own, cons.N, cons.Cout))
    for op in reversed(self.ops):
      op.plan_backward()
    return self

  def _plan_x_records(self, lib):
    """ResnetBlockBigGANpp with a shortcut: GroupNorm_0 and the 1x1 Conv_2 read the same (two-source) block input
    (layerspp.py:256, 283).  Conv_2 takes it as fp32 operands of the split kernel and would measure |x1|, |x2| first (two
    passes); GroupNorm_0's one-pass forward -- which runs first and holds those values in registers -- leaves the maxima
    in Conv_2's amax buffer instead (stk_gn_fwd_pl_max_f32 / stk_conv2d_fwd_rec_f32; the weight gradient reuses them)."""
    if os.environ.get('STK_X_RECORDS', '1') == '0' or not hasattr(lib, 'gn_fwd_pl_max_f32'):
      return
    index = {id(op): i for i, op in enumerate(self.ops)}
    found = False
    for op in self.ops:
      if not isinstance(op, Conv) or op.pl_fwd or op.x_from is not None or not op._kind(lib, 'fwd').endswith('.x2'):
        continue
      for gn in self.ops:
        if (isinstance(gn, GroupNormAct) and gn.x1 is op.x1 and gn.x2 is op.x2 and gn.fused and gn.y.pl_maker is gn and
            gn.xmax_for is None and index[id(gn)] < index[id(op)]):
          gn.xmax_for, op.x_from = op, gn
          found = True
          break
    if found:
      self.ops.insert(0, ZeroXRecords(self.amax_block, len(self.conv_amax)))

  def _grad_writers(self):
    """act tensor id -> ops that write its gradient, in FORWARD order (so [0] is the last writer of the backward)."""
    w = {}
    for op in self.ops:
      ins = op.inputs
      if isinstance(op, Conv) and (op.dy_peer is not None or op.res_via is not None):
        ins = (op.x1, op.x2)
      seen = set()
      for t in ins:
        if t is not None and t.space == 'act' and t.needs_grad and id(t) not in seen:
          seen.add(id(t))
          w.setdefault(id(t), []).append(op)
    return w

  def _plan_res_via(self, lib):
    """ResnetBlockBigGANpp without a shortcut convolution: out = (x + Conv_1(h)) / sqrt 2 with h = ... GroupNorm_0(x) ...
    (layerspp.py:256-287) is planned as Conv_1 with res = x.  d(x) receives d(out) / sqrt 2 from the skip and the
    GroupNorm_0 gradient; instead of Conv_1's backward writing the first in a pass of its own (read d(out), read-modify-
    write d(x)) the GroupNorm backward -- which writes d(x) anyway -- adds it on the way (stk_gn_bwd_out_f32 dx1_add)."""
    if os.environ.get('STK_RES_VIA', '1') == '0':
      return
    index = {id(op): i for i, op in enumerate(self.ops)}
    for op in self.ops:
      if not isinstance(op, Conv) or op.res is None or not op.res.needs_grad or op.dy_peer is not None:
        continue
      r = op.res
      if r.shape != op.y.shape or r is op.x1 or r is op.x2 or op.y is self.output:
        continue
      for gn in self.ops:
        if (isinstance(gn, GroupNormAct) and gn.x1 is r and gn.add_from is None and index[id(gn)] < index[id(op)] and
            int(lib.gn_bwd_out_ok(gn.C1, gn.C2, gn.HW, gn.G))):
          gn.add_from, op.res_via = op, gn
          break

  def _plan_dy_producers(self, lib):
    """Who writes a convolution's output gradient LAST?  For the 3x3 convolutions of a ResnetBlockBigGANpp it is a
    GroupNorm backward: Conv_0's output is read only by GroupNorm_1 (layerspp.py:273-278), and a block's output (Conv_1's)
    is first read by the next block's GroupNorm_0, whose backward therefore runs after every other contribution to
    d(out) has been accumulated.  That register-resident kernel sums the final values it stores per (sample, channel) and
    takes their maximum (stk_gn_bwd_out_f32), so the convolution's backward needs no pass over dy for its bias / time-
    embedding gradients and its planes' scale record (it went: read dy for sums and maxima, read it again to split) --
    nor does its shortcut peer (_plan_shared_dy), which shares the record and the fold entry."""
    if os.environ.get('STK_DY_PRODUCER', '1') == '0':
      return
    writers = self._grad_writers()
    for op in self.ops:
      if not isinstance(op, Conv) or op.Cout > 256:
        continue
      # consumers: the plane-operand 3x3 layers, and the fp32-operand layers on the two-way split (NIN_3 of the attention
      # blocks), whose data / weight gradient calls take the record in their own amax buffer (stk_conv2d_dgrad_rec_f32)
      if not (op.pl_dgrad or op.pl_wgrad) and not (op.dy_peer is None and op._kind(lib, 'dgrad').endswith('.x2')):
        continue
      if op.dy_from is not None or op.y is self.output or not op.y.needs_grad:
        continue
      if op.res is not None and op.res.needs_grad and op.dy_peer is None and op.res_via is None:
        continue                                   # its pass over dy also writes d(res): nothing to save
      if op.bias is None and (op.temb is None or not op.temb.needs_grad):
        continue
      if op.pl_dgrad and op.w.needs_grad and not op.pl_wgrad and op.dy_peer is not None:
        continue                                   # an fp32-operand weight gradient looks for the record in op.amax, not the peer's
      ws = writers.get(id(op.y), [])
      if not ws or not isinstance(ws[0], GroupNormAct):
        continue
      gn = ws[0]
      if gn.x1 is not op.y or gn.x2 is op.y or gn.dy_cons is not None:
        continue
      if not int(lib.gn_bwd_out_ok(gn.C1, gn.C2, gn.HW, gn.G)):
        continue
      gn.dy_cons, op.dy_prod = op, gn
    if any(isinstance(op, GroupNormAct) and op.dy_cons is not None for op in self.ops):
      self.ops.append(ZeroRecords(self.amax_block, len(self.conv_amax)))

  def _plan_shared_dy(self, lib):
    """ResnetBlockBigGANpp with a shortcut convolution: out = (Conv_2(x) + Conv_1(h)) / sqrt 2 (layerspp.py:283-287) is
    planned as Conv_1 with res = Conv_2's output.  That output has one reader, so its gradient is just dy(Conv_1) /
    out_div: instead of writing it (one pass), measuring it (another) and summing it for Conv_2's bias (a third), Conv_1's
    own pass over dy serves both layers and Conv_2 differentiates from dy(Conv_1) directly."""
    if os.environ.get('STK_SHARED_DY', '1') == '0' or not hasattr(lib, 'bias_grad_amax_dual_f32'):
      return
    readers = {}
    for op in self.ops:
      for v in vars(op).values():
        if isinstance(v, Tensor) and v.space == 'act' and v.producer is not op:
          readers.setdefault(id(v), []).append(op)
    for op in self.ops:
      if not isinstance(op, Conv) or op.res is None or not op.res.needs_grad:
        continue
      r = op.res
      P = r.producer
      if not isinstance(P, Conv) or P is op or r is self.output or readers.get(id(r), []) != [op]:
        continue
      if r is op.x1 or r is op.x2 or op.temb is r:
        continue
      # the consumer must take the one-pass bias / record path (maps below 64 x 64, <= 256 channels, a bias to sum),
      # and the shortcut must have nothing in its own backward that the pass cannot provide
      if not ((op.pl_dgrad or op.pl_wgrad) and op.Cout <= 256 and op.OH * op.OW < 4096 and op.bias is not None):
        continue
      if P.temb is not None or P.res is not None or P.bias is None or P.dy_from is not None or P.dy_peer is not None:
        continue
      if P.pl_dgrad or P.pl_wgrad or P.Cout != op.Cout:
        continue
      op.dy_peer, P.dy_from = P, op
      if ((P.x1.needs_grad or (P.x2 is not None and P.x2.needs_grad)) and
          hasattr(lib, 'conv2d_pl_ok') and
          int(lib.conv2d_pl_ok(1, P.C1, P.C2, P.N, P.H, P.W, P.Cout, P.KH, P.KW, 1, P.pad))):
        if op.dypl_off is not None:
          P.peer_planes = 'own'
        elif not self.own_dypl and self.ops.index(P) + 1 == self.ops.index(op):
          P.peer_planes = 'scratch'

  def _plan_f32_copies(self, lib):
    """For every GroupNorm output that is made as planes: who still reads its fp32 NCHW copy?  A convolution whose
    forward takes planes does not; its weight gradient does unless it takes planes too.  Anything else does."""
    readers = {}
    for op in self.ops:
      for v in vars(op).values():
        if isinstance(v, Tensor) and v.space == 'act' and v.producer is not op:
          readers.setdefault(id(v), []).append(op)
    for op in self.ops:
      if not isinstance(op, GroupNormAct) or op.y.pl_maker is not op:
        continue
      op.fused = bool(int(lib.gn_fwd_pl_fused(op.C1, op.C2, op.HW, op.G))) if hasattr(lib, 'gn_fwd_pl_fused') else False
      y = op.y
      fwd = bwd = y is self.output
      for r in readers.get(id(y), []):
        as_x1 = isinstance(r, Conv) and r.x1 is y and r.x2 is not y and r.res is not y and r.temb is not y
        if not (as_x1 and r.pl_fwd):
          fwd = True
        if not as_x1 or (r.w.needs_grad and not r.pl_wgrad) or not r.pl_fwd:
          bwd = True
      y.f32_fwd, y.f32_bwd = fwd, bwd
