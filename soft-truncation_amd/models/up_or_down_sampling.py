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
    with stk_lib.device_guard(x.device):
      lib.resample_naive_f32(x.data_ptr(), out.data_ptr(), N * C, H, W, 0 if up else 1, 1.0, 0.0,
                             stk_lib.stream_ptr(x.device))
    return out

  @staticmethod
  def backward(ctx, g):
    lib = _backend.get()
    g = g.contiguous()
    N, C, H, W = ctx.shape
    gx = torch.empty(ctx.shape, dtype=g.dtype, device=g.device)
    with stk_lib.device_guard(g.device):
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


class _Conv2dNoPad(torch.autograd.Function):
  """y = conv2d(x, w, stride, padding 0) on the C-ABI convolution kernels (forward, data gradient, weight gradient);
  the tensor-level counterpart of engine.graph.Conv for callers outside the planned graph."""

  @staticmethod
  def _dims(x, w, stride):
    N, C, H, W = x.shape
    Cout, _, KH, KW = w.shape
    return N, H, W, Cout, (H - KH) // stride + 1, (W - KW) // stride + 1, KH, KW, stride, 0

  @staticmethod
  def forward(ctx, x, w, stride):
    lib = _backend.get()
    _backend.check(x, lib)
    x, w = x.contiguous(), w.contiguous()
    dims = _Conv2dNoPad._dims(x, w, stride)
    y = torch.empty((dims[0], dims[3], dims[4], dims[5]), dtype=x.dtype, device=x.device)
    with stk_lib.device_guard(x.device):
      lib.conv2d_fwd_f32(x.data_ptr(), x.shape[1], None, 0, w.data_ptr(), 0, None, None, 0, None, 1.0, y.data_ptr(),
                         *dims, None, 0, stk_lib.stream_ptr(x.device))
    ctx.save_for_backward(x, w)
    ctx.stride = stride
    return y

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, gy):
    lib = _backend.get()
    x, w = ctx.saved_tensors
    gy = gy.contiguous()
    dims = _Conv2dNoPad._dims(x, w, ctx.stride)
    C = x.shape[1]
    gx = gw = None
    with stk_lib.device_guard(x.device):
      stream = stk_lib.stream_ptr(x.device)
      if ctx.needs_input_grad[0]:
        gx = torch.empty_like(x)
        lib.conv2d_dgrad_f32(gy.data_ptr(), w.data_ptr(), 0, gx.data_ptr(), C, 0.0, None, 0, 0.0, 1.0, *dims,
                             None, 0, stream)
      if ctx.needs_input_grad[1]:
        gw = torch.zeros_like(w)
        nb = int(lib.conv2d_wgrad_ws_bytes(C, 0, dims[0], dims[3], dims[4], dims[5], dims[6], dims[7]))
        ws = torch.empty(max(nb // 4, 64), dtype=torch.float32, device=x.device)
        lib.conv2d_wgrad_f32(x.data_ptr(), C, None, 0, gy.data_ptr(), gw.data_ptr(), 0, 1.0, ws.data_ptr(), nb, *dims,
                             stream)
    return gx, gw, None


def conv_downsample_2d(x, w, k=None, factor=2, gain=1):
  """Fused ``downsample_2d`` + ``conv2d`` (models/up_or_down_sampling.py:144-178): FIR pre-filter with padding
  ((p+1)//2, p//2), p = (len(k) - factor) + (conv_w - 1), then the convolution with stride `factor`, padding 0.
  x [N, C, H, W], w [Cout, C, kh, kw] (kh == kw)."""
  assert isinstance(factor, int) and factor >= 1
  _, _, conv_h, conv_w = w.shape
  assert conv_w == conv_h
  taps, pad = _conv_down_taps(k, factor, gain, conv_w)
  x = upfirdn2d(x, torch.tensor(taps, device=x.device), pad=pad)
  return _Conv2dNoPad.apply(x, w, factor)


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
