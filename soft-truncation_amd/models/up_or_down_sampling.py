"""StyleGAN2-style FIR resampling on the HIP ``upfirdn2d`` kernel
(reference: models/up_or_down_sampling.py).

Tensor-level functions (``upsample_2d`` :195-224, ``downsample_2d`` :227-257,
``conv_downsample_2d`` :144-178, ``naive_upsample_2d`` :59-63, ``naive_downsample_2d`` :66-69)
keep the reference signatures and run on ``op.upfirdn2d`` / the C-ABI kernels.  The ``emit_*``
functions put the same operators into the engine's planned graph, which is what the score
network uses.

``upsample_conv_2d`` (:72-141) is dead in the reference (``w[..., ::-1, ::-1]`` raises in
PyTorch), so it raises here as well.
"""
import numpy as np
import torch
import torch.nn as nn

from ..engine import graph as G
from ..engine import lib as stk_lib
from ..op import upfirdn2d
from ..op import _backend


def _setup_kernel(k):
  k = np.asarray(k, dtype=np.float32)
  if k.ndim == 1:
    k = np.outer(k, k)
  k /= np.sum(k)
  assert k.ndim == 2
  assert k.shape[0] == k.shape[1]
  return k


def _shape(x, dim):
  return x.shape[dim]


def _up_taps(k, factor, gain):
  if k is None:
    k = [1] * factor
  k = _setup_kernel(k) * (gain * (factor ** 2))
  p = k.shape[0] - factor
  return k, ((p + 1) // 2 + factor - 1, p // 2)


def _down_taps(k, factor, gain):
  if k is None:
    k = [1] * factor
  k = _setup_kernel(k) * gain
  p = k.shape[0] - factor
  return k, ((p + 1) // 2, p // 2)


def _conv_down_taps(k, factor, gain, conv_w):
  if k is None:
    k = [1] * factor
  k = _setup_kernel(k) * gain
  p = (k.shape[0] - factor) + (conv_w - 1)
  return k, ((p + 1) // 2, p // 2)


# ---------------------------------------------------------------------------------------------
# tensor-level API (reference signatures)
# ---------------------------------------------------------------------------------------------
def upsample_2d(x, k=None, factor=2, gain=1):
  assert isinstance(factor, int) and factor >= 1
  taps, pad = _up_taps(k, factor, gain)
  return upfirdn2d(x, torch.tensor(taps, device=x.device), up=factor, pad=pad)


def downsample_2d(x, k=None, factor=2, gain=1):
  assert isinstance(factor, int) and factor >= 1
  taps, pad = _down_taps(k, factor, gain)
  return upfirdn2d(x, torch.tensor(taps, device=x.device), down=factor, pad=pad)


class _NaiveResample(torch.autograd.Function):
  @staticmethod
  def forward(ctx, x, up):
    lib = _backend.get()
    _backend.check(x, lib)
    x = x.contiguous()
    N, C, H, W = x.shape
    ctx.up, ctx.shape = up, (N, C, H, W)
    out = torch.empty((N, C, H * 2, W * 2) if up else (N, C, H // 2, W // 2), dtype=x.dtype, device=x.device)
    lib.resample_naive_f32(x.data_ptr(), out.data_ptr(), N * C, H, W, 0 if up else 1, 1.0, 0.0,
                           stk_lib.stream_ptr(x.device))
    return out

  @staticmethod
  def backward(ctx, g):
    lib = _backend.get()
    g = g.contiguous()
    N, C, H, W = ctx.shape
    gx = torch.empty(ctx.shape, dtype=g.dtype, device=g.device)
    if ctx.up:
      lib.resample_naive_f32(g.data_ptr(), gx.data_ptr(), N * C, 2 * H, 2 * W, 1, 4.0, 0.0,
                             stk_lib.stream_ptr(g.device))
    else:
      lib.resample_naive_f32(g.data_ptr(), gx.data_ptr(), N * C, H // 2, W // 2, 0, 0.25, 0.0,
                             stk_lib.stream_ptr(g.device))
    return gx, None


def naive_upsample_2d(x, factor=2):
  assert factor == 2, 'the kernel implements the factor-2 case used by every config'
  return _NaiveResample.apply(x, True)


def naive_downsample_2d(x, factor=2):
  assert factor == 2, 'the kernel implements the factor-2 case used by every config'
  return _NaiveResample.apply(x, False)


def upsample_conv_2d(x, w, k=None, factor=2, gain=1):
  raise NotImplementedError('upsample_conv_2d is dead code in the reference '
                            '(models/up_or_down_sampling.py:126 raises); not provided')


# ---------------------------------------------------------------------------------------------
# graph emitters
# ---------------------------------------------------------------------------------------------
def emit_upsample_2d(g, x, k=None, factor=2, gain=1, name='fir_up'):
  taps, pad = _up_taps(k, factor, gain)
  return g.add(G.UpFirDn(g, x, taps, factor, 1, pad, name))


def emit_downsample_2d(g, x, k=None, factor=2, gain=1, name='fir_down'):
  taps, pad = _down_taps(k, factor, gain)
  return g.add(G.UpFirDn(g, x, taps, 1, factor, pad, name))


def emit_naive_upsample_2d(g, x, name='up'):
  return g.add(G.ResampleNaive(g, x, True, name))


def emit_naive_downsample_2d(g, x, name='down'):
  return g.add(G.ResampleNaive(g, x, False, name))


class Conv2d(nn.Module):
  """Conv2d with fused FIR up/down-sampling (models/up_or_down_sampling.py:23-56)."""

  def __init__(self, in_ch, out_ch, kernel, up=False, down=False, resample_kernel=(1, 3, 3, 1),
               use_bias=True, kernel_init=None):
    super().__init__()
    assert not (up and down)
    assert kernel >= 1 and kernel % 2 == 1
    self.weight = nn.Parameter(torch.zeros(out_ch, in_ch, kernel, kernel))
    if kernel_init is not None:
      self.weight.data = kernel_init(self.weight.data.shape)
    if use_bias:
      self.bias = nn.Parameter(torch.zeros(out_ch))
    self.up = up
    self.down = down
    self.resample_kernel = resample_kernel
    self.kernel = kernel
    self.use_bias = use_bias

  def emit(self, g, x, res=None, out_div=1.0, name='uds_conv'):
    bias = self.bias if self.use_bias else None
    if self.up:
      raise NotImplementedError('Conv2d(up=True) goes through upsample_conv_2d, which is dead in the reference')
    if self.down:
      # conv_downsample_2d (:144-178): FIR with pad ((p+1)//2, p//2) then stride-2 conv, padding 0
      taps, pad = _conv_down_taps(self.resample_kernel, 2, 1, self.kernel)
      h = g.add(G.UpFirDn(g, x, taps, 1, 1, pad, name + '.fir'))
      return g.conv(h, None, self.weight, bias, w_layout=0, stride=2, pad=0, res=res, out_div=out_div, name=name)
    return g.conv(x, None, self.weight, bias, w_layout=0, stride=1, pad=self.kernel // 2, res=res,
                  out_div=out_div, name=name)
