"""`op.upfirdn2d` -- FIR up/down-sampling (interface of reference op/upfirdn2d.py:145-156, NCHW input,
`upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0))`).

upfirdn2d is a LINEAR map of its input, and its transpose is again an upfirdn2d: the flipped taps, up and down
swapped, and the paddings of op/upfirdn2d.py:111-114.  The autograd side is built on exactly that and nothing else:
one :class:`FirMap` value describes a member of the family (factors, paddings, input and output extent) and knows its
transpose; one autograd function applies a `FirMap`, and its backward applies the transposed `FirMap` through the
same function.  Differentiating any number of times therefore needs no further code (the reference spells out a second
Function for the double backward, :62-85; here `transpose(transpose(m))` is `m` again, up to padding no output reads).

The work is done by ``stk_upfirdn2d_f32`` (hand-written HIP for gfx950, csrc/upfirdn2d.hip; ``_f16`` / ``_f64`` for half
and double tensors, the other types the reference's extension dispatches on) on torch's current HIP stream.  The reference JIT-builds a CUDA extension at import and silently runs a PyTorch implementation for CPU
tensors (:145-156); here a tensor that does not live on the GPU is an error, never a fallback.
"""
import collections

import torch

from ..engine import lib as stk_lib
from . import _backend


class FirMap(collections.namedtuple('FirMap', 'up down pad in_hw out_hw')):
  """One upfirdn2d operator on planes: per-axis factors ``up = (x, y)``, ``down = (x, y)``, paddings
  ``pad = (x0, x1, y0, y1)``, and the plane extents it maps between."""
  __slots__ = ()

  @staticmethod
  def of(in_hw, taps_hw, up, down, pad):
    (h, w), (kh, kw) = in_hw, taps_hw
    out_h = (h * up[1] + pad[2] + pad[3] - kh) // down[1] + 1
    out_w = (w * up[0] + pad[0] + pad[1] - kw) // down[0] + 1
    return FirMap(tuple(up), tuple(down), tuple(pad), (h, w), (out_h, out_w))

  def transpose(self, taps_hw):
    """The adjoint operator (applied with the flipped taps): out-extent planes -> in-extent planes."""
    (h, w), (oh, ow), (kh, kw) = self.in_hw, self.out_hw, taps_hw
    (ux, uy), (dx, dy), (px0, _, py0, _) = self.up, self.down, self.pad
    pad_t = (kw - px0 - 1, w * ux - ow * dx + px0 - ux + 1,
             kh - py0 - 1, h * uy - oh * dy + py0 - uy + 1)
    return FirMap(self.down, self.up, pad_t, self.out_hw, self.in_hw)


# the reference's pybind function dispatches on every floating type and half (op/upfirdn2d_kernel.cu:311)
_ENTRY = {torch.float32: 'upfirdn2d_f32', torch.float16: 'upfirdn2d_f16', torch.float64: 'upfirdn2d_f64'}


def _run(planes, taps, m):
  """planes [P, h, w] contiguous (float32 / float16 / float64) on the device -> [P, oh, ow]; taps of the same type."""
  lib = _backend.get()
  _backend.check(planes, lib)
  if planes.dtype not in _ENTRY:
    raise TypeError(f'upfirdn2d: {planes.dtype} tensors are not supported (float32, float16, float64 are)')
  out = torch.empty((planes.shape[0],) + m.out_hw, dtype=planes.dtype, device=planes.device)
  with stk_lib.device_guard(planes.device):
    getattr(lib, _ENTRY[planes.dtype])(planes.data_ptr(), taps.data_ptr(), out.data_ptr(), planes.shape[0], m.in_hw[0], m.in_hw[1], 1,
                      taps.shape[0], taps.shape[1], m.up[0], m.up[1], m.down[0], m.down[1],
                      m.pad[0], m.pad[1], m.pad[2], m.pad[3], stk_lib.stream_ptr(planes.device))
  return out


class _ApplyFir(torch.autograd.Function):
  """y = M x for a FirMap M; dL/dx = M^T dL/dy, computed by this same function."""

  @staticmethod
  def forward(ctx, x, taps, m):
    lead = x.shape[:-2]
    assert tuple(x.shape[-2:]) == m.in_hw, (x.shape, m)
    ctx.m = m
    ctx.save_for_backward(taps)
    y = _run(x.reshape((-1,) + m.in_hw).contiguous(), taps, m)
    return y.view(lead + m.out_hw)

  @staticmethod
  def backward(ctx, gy):
    taps, = ctx.saved_tensors
    mt = ctx.m.transpose(tuple(taps.shape))
    return _ApplyFir.apply(gy, torch.flip(taps, (0, 1)).contiguous(), mt), None, None


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
  """[B, C, H, W] -> [B, C, H', W']: zero-insert by `up`, pad by `pad = (before, after)` on both axes (negative =
  crop), correlate with the flipped `kernel`, keep every `down`-th sample (reference op/upfirdn2d.py:159-200)."""
  taps = kernel.detach().to(device=input.device, dtype=input.dtype).contiguous()
  m = FirMap.of(tuple(input.shape[-2:]), tuple(taps.shape), (up, up), (down, down), (pad[0], pad[1], pad[0], pad[1]))
  return _ApplyFir.apply(input, taps, m)
