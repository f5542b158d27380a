"""`op.fused_leaky_relu` / `op.FusedLeakyReLU` (interface of reference op/fused_act.py:75-97):
``out = leaky_relu(input + bias[channel], negative_slope) * scale`` on ``stk_fused_bias_act_f32``
(csrc/elementwise.hip, the drop-in for the reference's `fused_bias_act` pybind function).

Autograd is organised around the one fact that makes this op cheap to differentiate: with the sign pattern of the
forward OUTPUT frozen, everything downstream of it is the linear map

    gate_y(v, c) = scale * (v + c[channel]) * (y > 0 ? 1 : negative_slope)            (kernel mode act 3 / grad 1)

The forward is the one nonlinear launch (mode act 3 / grad 0) and keeps its output y; its vector-Jacobian product is
``gate_y(g, 0)`` for the input and the per-channel sum of that for the bias; and `gate_y` -- being linear in (v, c) --
is its own derivative, so gradients of gradients (op/fused_act.py:44-52 in the reference) come from the same function.
Nothing in the live NCSN++ graph calls this op; it is kept because it is part of the reference's
native-op surface.  The reference's CPU branch ignores ``negative_slope`` (hard-coded 0.2, :91); the kernel path --
the only one here, CPU tensors are an error -- honours it.
"""
import torch
from torch import nn

from ..engine import lib as stk_lib
from . import _backend

_LRELU = 3     # `act` code of the kernel (fused_bias_act_kernel.cu:36-47)
# the reference's pybind function dispatches on every floating type and half (op/fused_bias_act_kernel.cu:77)
_ENTRY = {torch.float32: 'fused_bias_act_f32', torch.float16: 'fused_bias_act_f16', torch.float64: 'fused_bias_act_f64'}


def _launch(x, bias, y_ref, through_ref, slope, scale):
  lib = _backend.get()
  _backend.check(x, lib)
  if x.dtype not in _ENTRY:
    raise TypeError(f'fused_leaky_relu: {x.dtype} tensors are not supported (float32, float16, float64 are)')
  x = x.contiguous()
  out = torch.empty_like(x)
  inner = 1
  for d in x.shape[2:]:
    inner *= d
  b = bias.to(x.dtype).contiguous() if bias is not None else None
  r = y_ref.to(x.dtype).contiguous() if y_ref is not None else None
  with stk_lib.device_guard(x.device):
    getattr(lib, _ENTRY[x.dtype])(x.data_ptr(), b.data_ptr() if b is not None else None,
                           r.data_ptr() if r is not None else None, out.data_ptr(), x.numel(), inner,
                           b.numel() if b is not None else 1, _LRELU, 1 if through_ref else 0,
                           float(slope), float(scale), stk_lib.stream_ptr(x.device))
  return out


def _per_channel_sum(t):
  return t.sum([d for d in range(t.dim()) if d != 1])


class _Gate(torch.autograd.Function):
  """gate_y(v, c): linear in v and c, so its backward is itself."""

  @staticmethod
  def forward(ctx, v, c, y, slope, scale):
    ctx.save_for_backward(y)
    ctx.cfg = (slope, scale, c is not None)
    return _launch(v, c, y, True, slope, scale)

  @staticmethod
  def backward(ctx, g):
    y, = ctx.saved_tensors
    slope, scale, has_c = ctx.cfg
    gv = _Gate.apply(g, None, y, slope, scale)
    return gv, (_per_channel_sum(gv) if has_c else None), None, None, None


class _BiasLeakyReLU(torch.autograd.Function):
  @staticmethod
  def forward(ctx, x, bias, slope, scale):
    y = _launch(x, bias, None, False, slope, scale)
    ctx.save_for_backward(y)
    ctx.cfg = (slope, scale)
    return y

  @staticmethod
  def backward(ctx, g):
    y, = ctx.saved_tensors
    gx = _Gate.apply(g, None, y, *ctx.cfg)
    return gx, _per_channel_sum(gx), None, None


def fused_leaky_relu(input, bias, negative_slope=0.2, scale=2 ** 0.5):
  return _BiasLeakyReLU.apply(input, bias, negative_slope, scale)


class FusedLeakyReLU(nn.Module):
  """Per-channel learnable bias + leaky ReLU + gain (state_dict key ``bias``, as the reference's module)."""

  def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
    super().__init__()
    self.bias = nn.Parameter(torch.zeros(channel))
    self.negative_slope, self.scale = negative_slope, scale

  def forward(self, input):
    return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)
