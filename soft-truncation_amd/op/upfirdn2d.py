"""`op.upfirdn2d` -- FIR up/down-sampling (reference: op/upfirdn2d.py:19-156).

Same call signature as the reference (`upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0))` on
an NCHW tensor) and the same autograd structure: the backward of the operator is the operator
itself with the flipped taps, up and down swapped and the ``g_pad`` paddings
(op/upfirdn2d.py:111-114), and that backward is itself differentiable (:62-85).

The reference JIT-builds a CUDA extension at import and silently falls back to a PyTorch
implementation for CPU tensors (:145-156).  Here the work is done by
``stk_upfirdn2d_f32`` (hand-written HIP for gfx950, csrc/upfirdn2d.hip) on the current HIP
stream; a tensor that does not live on the GPU is an error, never a fallback.
"""
import torch
from torch.autograd import Function

from ..engine import lib as stk_lib
from . import _backend


def _launch(inp, kernel, out_hw, up, down, pad):
  """inp: [major, in_h, in_w, 1] contiguous.  Returns [major, out_h, out_w, 1]."""
  lib = _backend.get()
  _backend.check(inp, lib)
  inp = inp.contiguous()
  kernel = kernel.contiguous().to(device=inp.device, dtype=torch.float32)
  major, in_h, in_w, minor = inp.shape
  out = torch.empty((major, out_hw[0], out_hw[1], minor), dtype=inp.dtype, device=inp.device)
  lib.upfirdn2d_f32(inp.data_ptr(), kernel.data_ptr(), out.data_ptr(), major, in_h, in_w, minor,
                    kernel.shape[0], kernel.shape[1], up[0], up[1], down[0], down[1],
                    pad[0], pad[1], pad[2], pad[3], stk_lib.stream_ptr(inp.device))
  return out


class UpFirDn2dBackward(Function):
  @staticmethod
  def forward(ctx, grad_output, kernel, grad_kernel, up, down, pad, g_pad, in_size, out_size):
    grad_output = grad_output.reshape(-1, out_size[0], out_size[1], 1)
    grad_input = _launch(grad_output, grad_kernel, (in_size[2], in_size[3]), down, up, g_pad)
    grad_input = grad_input.view(in_size[0], in_size[1], in_size[2], in_size[3])
    ctx.save_for_backward(kernel)
    ctx.up, ctx.down, ctx.pad = up, down, pad
    ctx.in_size, ctx.out_size = in_size, out_size
    return grad_input

  @staticmethod
  def backward(ctx, gradgrad_input):
    kernel, = ctx.saved_tensors
    gradgrad_input = gradgrad_input.reshape(-1, ctx.in_size[2], ctx.in_size[3], 1)
    gradgrad_out = _launch(gradgrad_input, kernel, ctx.out_size, ctx.up, ctx.down, ctx.pad)
    gradgrad_out = gradgrad_out.view(ctx.in_size[0], ctx.in_size[1], ctx.out_size[0], ctx.out_size[1])
    return gradgrad_out, None, None, None, None, None, None, None, None


class UpFirDn2d(Function):
  @staticmethod
  def forward(ctx, input, kernel, up, down, pad):
    up_x, up_y = up
    down_x, down_y = down
    pad_x0, pad_x1, pad_y0, pad_y1 = pad
    kernel_h, kernel_w = kernel.shape
    batch, channel, in_h, in_w = input.shape
    ctx.in_size = input.shape
    out_h = (in_h * up_y + pad_y0 + pad_y1 - kernel_h) // down_y + 1
    out_w = (in_w * up_x + pad_x0 + pad_x1 - kernel_w) // down_x + 1
    ctx.out_size = (out_h, out_w)
    ctx.up, ctx.down, ctx.pad = (up_x, up_y), (down_x, down_y), (pad_x0, pad_x1, pad_y0, pad_y1)
    ctx.g_pad = (kernel_w - pad_x0 - 1,
                 in_w * up_x - out_w * down_x + pad_x0 - up_x + 1,
                 kernel_h - pad_y0 - 1,
                 in_h * up_y - out_h * down_y + pad_y0 - up_y + 1)
    ctx.save_for_backward(kernel, torch.flip(kernel, [0, 1]))
    out = _launch(input.reshape(-1, in_h, in_w, 1), kernel, ctx.out_size, ctx.up, ctx.down, ctx.pad)
    return out.view(-1, channel, out_h, out_w)

  @staticmethod
  def backward(ctx, grad_output):
    kernel, grad_kernel = ctx.saved_tensors
    grad_input = UpFirDn2dBackward.apply(grad_output, kernel, grad_kernel, ctx.up, ctx.down, ctx.pad,
                                         ctx.g_pad, ctx.in_size, ctx.out_size)
    return grad_input, None, None, None, None


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
  return UpFirDn2d.apply(input, kernel, (up, up), (down, down), (pad[0], pad[1], pad[0], pad[1]))
