"""`op.fused_leaky_relu` / `op.FusedLeakyReLU` (reference: op/fused_act.py:20-97).

out = leaky_relu(input + bias[channel], negative_slope) * scale, with the reference's
forward / grad / grad-grad structure on top of ``stk_fused_bias_act_f32`` (csrc/elementwise.hip).
Nothing in the live NCSN++ graph calls it; it is kept because it is part of the
reference's native-op surface.  Note the reference's CPU branch ignores ``negative_slope`` and
hard-codes 0.2 (op/fused_act.py:91); the kernel path -- the only one here -- honours it.
"""
import torch
from torch import nn
from torch.autograd import Function

from ..engine import lib as stk_lib
from . import _backend


def _bias_act(x, bias, ref, act, grad, alpha, scale):
  lib = _backend.get()
  _backend.check(x, lib)
  x = x.contiguous()
  out = torch.empty_like(x)
  use_b = bias is not None and bias.numel() > 0
  use_r = ref is not None and ref.numel() > 0
  step_b = 1
  for i in range(2, x.dim()):
    step_b *= x.size(i)
  b = bias.contiguous() if use_b else None
  r = ref.contiguous() if use_r else None
  lib.fused_bias_act_f32(x.data_ptr(), b.data_ptr() if use_b else None, r.data_ptr() if use_r else None,
                         out.data_ptr(), x.numel(), step_b, b.numel() if use_b else 1, act, grad,
                         float(alpha), float(scale), stk_lib.stream_ptr(x.device))
  return out


class FusedLeakyReLUFunctionBackward(Function):
  @staticmethod
  def forward(ctx, grad_output, out, negative_slope, scale):
    ctx.save_for_backward(out)
    ctx.negative_slope = negative_slope
    ctx.scale = scale
    grad_input = _bias_act(grad_output, None, out, 3, 1, negative_slope, scale)
    dim = [0]
    if grad_input.ndim > 2:
      dim += list(range(2, grad_input.ndim))
    grad_bias = grad_input.sum(dim).detach()
    return grad_input, grad_bias

  @staticmethod
  def backward(ctx, gradgrad_input, gradgrad_bias):
    out, = ctx.saved_tensors
    gradgrad_out = _bias_act(gradgrad_input, gradgrad_bias, out, 3, 1, ctx.negative_slope, ctx.scale)
    return gradgrad_out, None, None, None


class FusedLeakyReLUFunction(Function):
  @staticmethod
  def forward(ctx, input, bias, negative_slope, scale):
    out = _bias_act(input, bias, None, 3, 0, negative_slope, scale)
    ctx.save_for_backward(out)
    ctx.negative_slope = negative_slope
    ctx.scale = scale
    return out

  @staticmethod
  def backward(ctx, grad_output):
    out, = ctx.saved_tensors
    grad_input, grad_bias = FusedLeakyReLUFunctionBackward.apply(grad_output, out, ctx.negative_slope, ctx.scale)
    return grad_input, grad_bias, None, None


class FusedLeakyReLU(nn.Module):
  def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
    super().__init__()
    self.bias = nn.Parameter(torch.zeros(channel))
    self.negative_slope = negative_slope
    self.scale = scale

  def forward(self, input):
    return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)


def fused_leaky_relu(input, bias, negative_slope=0.2, scale=2 ** 0.5):
  return FusedLeakyReLUFunction.apply(input, bias, negative_slope, scale)
