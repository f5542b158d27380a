"""NCSN++ building blocks (reference: models/layerspp.py).

Each class owns its parameters under the reference's attribute names (``GroupNorm_0``,
``Conv_0``, ``Dense_0``, ``NIN_0`` ... so checkpoints are key-compatible) and lowers itself into
the engine graph with ``emit``: GroupNorm+SiLU(+dropout) is one fused kernel, the 3x3 convs carry
the bias / time-embedding / residual / (1/sqrt 2) epilogues, channel concatenation is never
materialised (GroupNorm and the convs read two sources), and the 46 ``Dense_0`` projections are
evaluated as one GEMM by the caller (``temb`` below is that shared [B, sum Cout] tensor plus this
block's column offset).
"""
import numpy as np
import os

import torch
import torch.nn as nn

from . import layers
from . import up_or_down_sampling as uds
from ..engine.graph import SQRT2

conv1x1 = layers.ddpm_conv1x1
conv3x3 = layers.ddpm_conv3x3
NIN = layers.NIN
default_init = layers.default_init


def _graph_only(name):
  raise RuntimeError(f'{name} is evaluated as part of the planned score-network graph (NCSNpp.forward); '
                     f'it has no eager PyTorch path')


class FixedFouriereProjection(nn.Module):
  """Fixed Fourier input features (models/layerspp.py:31-43); ``model.fourier_feature`` is False
  in every shipped config."""

  def forward(self, x):
    _graph_only('FixedFouriereProjection')


class GaussianFourierProjection(nn.Module):
  """Gaussian Fourier embedding of the noise level; frozen ``W`` (models/layerspp.py:45-54)."""

  def __init__(self, embedding_size=256, scale=1.0):
    super().__init__()
    self.W = nn.Parameter(torch.randn(embedding_size) * scale, requires_grad=False)

  def forward(self, x):
    _graph_only('GaussianFourierProjection')


class Combine(nn.Module):
  """Combine a skip branch: conv1x1(x) then cat / sum with y (models/layerspp.py:57-72)."""

  def __init__(self, dim1, dim2, method='cat'):
    super().__init__()
    self.Conv_0 = conv1x1(dim1, dim2)
    self.method = method

  def emit(self, g, x, y, name='combine'):
    if self.method == 'sum':
      return layers.conv_emit(g, self.Conv_0, x, res=y, name=name)
    if self.method == 'cat':
      from ..engine import graph as G
      return g.add(G.Concat(g, layers.conv_emit(g, self.Conv_0, x, name=name + '.conv'), y, name=name))
    raise ValueError(f'Method {self.method} not recognized.')

  def forward(self, x, y):
    _graph_only('Combine')


class AttnBlockpp(nn.Module):
  """Single-head self-attention over the H*W positions (models/layerspp.py:75-104)."""

  def __init__(self, channels, skip_rescale=False, init_scale=0.):
    super().__init__()
    self.GroupNorm_0 = nn.GroupNorm(num_groups=min(channels // 4, 32), num_channels=channels, eps=1e-6)
    self.NIN_0 = NIN(channels, channels)
    self.NIN_1 = NIN(channels, channels)
    self.NIN_2 = NIN(channels, channels)
    self.NIN_3 = NIN(channels, channels, init_scale=init_scale)
    self.skip_rescale = skip_rescale

  def qkv_params(self):
    """The three projections as one: weights to be interleaved column-wise ([in, 3 out]), biases back to back
    (engine/flat.py; NCSNpp._flat_groups asks for exactly this layout)."""
    return [self.NIN_0.W, self.NIN_1.W, self.NIN_2.W], [self.NIN_0.b, self.NIN_1.b, self.NIN_2.b]

  def emit(self, g, x, name='attn'):
    from ..engine.graph import AttentionCore, Tensor
    h = g.gn_act(x, None, self.GroupNorm_0, act=False, name=name + '.gn')
    ws, bs = self.qkv_params()
    C = self.NIN_0.W.shape[1]
    w_off = g.flat.cols_block(ws) if hasattr(g.flat, 'cols_block') else None
    b0 = g.param(bs[0])
    stacked = (w_off is not None and
               all(g.flat.offset_of(b)[0] == b0.off + i * C for i, b in enumerate(bs)))
    if stacked:
      # q, k, v = three NIN layers on the same input (layerspp.py:91-93) as ONE 1x1 convolution with 3 C output
      # channels: h is read once, and the backward is one data gradient and one weight gradient instead of three each
      w_all = Tensor((ws[0].shape[0], 3 * C), 'param', w_off, True, name + '.Wqkv')
      w_all.goff = w_off
      b_all = Tensor((3 * C,), 'param', b0.off, True, name + '.bqkv')
      b_all.goff = b0.off
      qkv = g.conv1x1_t(h, w_all, b_all, 3 * C, name=name + '.qkv')
      o = g.add(AttentionCore(g, None, None, None, name=name, qkv=qkv))
    else:
      q = self.NIN_0.emit(g, h, name=name + '.q')
      k = self.NIN_1.emit(g, h, name=name + '.k')
      v = self.NIN_2.emit(g, h, name=name + '.v')
      o = g.add(AttentionCore(g, q, k, v, name=name))
    return self.NIN_3.emit(g, o, res=x, out_div=SQRT2 if self.skip_rescale else 1.0, name=name + '.out')

  def forward(self, x):
    _graph_only('AttnBlockpp')


class Upsample(nn.Module):
  """models/layerspp.py:107-139."""

  def __init__(self, in_ch=None, out_ch=None, with_conv=False, fir=False, fir_kernel=(1, 3, 3, 1)):
    super().__init__()
    out_ch = out_ch if out_ch else in_ch
    if not fir:
      if with_conv:
        self.Conv_0 = conv3x3(in_ch, out_ch)
    else:
      if with_conv:
        self.Conv2d_0 = uds.Conv2d(in_ch, out_ch, kernel=3, up=True, resample_kernel=fir_kernel,
                                   use_bias=True, kernel_init=default_init())
    self.fir = fir
    self.with_conv = with_conv
    self.fir_kernel = fir_kernel
    self.out_ch = out_ch

  def emit(self, g, x, res=None, out_div=1.0, name='upsample'):
    if not self.fir:
      h = uds.emit_naive_upsample_2d(g, x, name=name + '.nn')   # F.interpolate(..., 'nearest') x2
      if self.with_conv:
        h = layers.conv_emit(g, self.Conv_0, h, res=res, out_div=out_div, name=name)
      return h
    if not self.with_conv:
      return uds.emit_upsample_2d(g, x, self.fir_kernel, factor=2, name=name)
    return self.Conv2d_0.emit(g, x, res=res, out_div=out_div, name=name)

  def forward(self, x):
    _graph_only('Upsample')


class Downsample(nn.Module):
  """models/layerspp.py:142-176."""

  def __init__(self, in_ch=None, out_ch=None, with_conv=False, fir=False, fir_kernel=(1, 3, 3, 1)):
    super().__init__()
    out_ch = out_ch if out_ch else in_ch
    if not fir:
      if with_conv:
        self.Conv_0 = conv3x3(in_ch, out_ch, stride=2, padding=0)
    else:
      if with_conv:
        self.Conv2d_0 = uds.Conv2d(in_ch, out_ch, kernel=3, down=True, resample_kernel=fir_kernel,
                                   use_bias=True, kernel_init=default_init())
    self.fir = fir
    self.fir_kernel = fir_kernel
    self.with_conv = with_conv
    self.out_ch = out_ch

  def emit(self, g, x, res=None, out_div=1.0, name='downsample'):
    if not self.fir:
      if self.with_conv:
        # F.pad(x, (0, 1, 0, 1)) then stride-2 3x3 conv, padding 0: the extra bottom/right row is
        # read as zero by the kernel's bounds check.
        H, W = x.shape[2], x.shape[3]
        return g.conv(x, None, self.Conv_0.weight, self.Conv_0.bias, w_layout=0, stride=2, pad=0,
                      out_hw=((H + 1 - 3) // 2 + 1, (W + 1 - 3) // 2 + 1), res=res, out_div=out_div, name=name)
      return uds.emit_naive_downsample_2d(g, x, name=name)   # avg_pool2d(2, 2)
    if not self.with_conv:
      return uds.emit_downsample_2d(g, x, self.fir_kernel, factor=2, name=name)
    return self.Conv2d_0.emit(g, x, res=res, out_div=out_div, name=name)

  def forward(self, x):
    _graph_only('Downsample')


class ResnetBlockDDPMpp(nn.Module):
  """DDPM residual block (models/layerspp.py:179-222)."""

  def __init__(self, act, in_ch, out_ch=None, temb_dim=None, conv_shortcut=False, dropout=0.1,
               skip_rescale=False, init_scale=0.):
    super().__init__()
    out_ch = out_ch if out_ch else in_ch
    self.GroupNorm_0 = nn.GroupNorm(num_groups=min(in_ch // 4, 32), num_channels=in_ch, eps=1e-6)
    self.Conv_0 = conv3x3(in_ch, out_ch)
    if temb_dim is not None:
      self.Dense_0 = nn.Linear(temb_dim, out_ch)
      self.Dense_0.weight.data = default_init()(self.Dense_0.weight.data.shape)
      nn.init.zeros_(self.Dense_0.bias)
    self.GroupNorm_1 = nn.GroupNorm(num_groups=min(out_ch // 4, 32), num_channels=out_ch, eps=1e-6)
    self.Dropout_0 = nn.Dropout(dropout)
    self.Conv_1 = conv3x3(out_ch, out_ch, init_scale=init_scale)
    if in_ch != out_ch:
      if conv_shortcut:
        self.Conv_2 = conv3x3(in_ch, out_ch)
      else:
        self.NIN_0 = NIN(in_ch, out_ch)
    self.skip_rescale = skip_rescale
    self.act = act
    self.in_ch = in_ch
    self.out_ch = out_ch
    self.conv_shortcut = conv_shortcut

  def emit(self, g, x1, x2, temb, temb_col, name='res'):
    h = g.gn_act(x1, x2, self.GroupNorm_0, act=True, name=name + '.gn0')
    h = layers.conv_emit(g, self.Conv_0, h, temb=temb, temb_col=temb_col, name=name + '.conv0')
    h = g.gn_act(h, None, self.GroupNorm_1, act=True, drop_p=self.Dropout_0.p, name=name + '.gn1')
    if self.in_ch != self.out_ch:
      if self.conv_shortcut:
        xs = layers.conv_emit(g, self.Conv_2, x1, x2, name=name + '.sc')
      else:
        xs = self.NIN_0.emit(g, x1, x2, name=name + '.sc')
    else:
      assert x2 is None
      xs = x1
    return layers.conv_emit(g, self.Conv_1, h, res=xs, out_div=SQRT2 if self.skip_rescale else 1.0,
                            name=name + '.conv1')

  def forward(self, x, temb=None):
    _graph_only('ResnetBlockDDPMpp')


class ResnetBlockBigGANpp(nn.Module):
  """BigGAN residual block with optional FIR / naive up- or down-sampling
  (models/layerspp.py:225-287)."""

  def __init__(self, act, in_ch, out_ch=None, temb_dim=None, up=False, down=False, dropout=0.1,
               fir=False, fir_kernel=(1, 3, 3, 1), skip_rescale=True, init_scale=0.):
    super().__init__()
    out_ch = out_ch if out_ch else in_ch
    self.GroupNorm_0 = nn.GroupNorm(num_groups=min(in_ch // 4, 32), num_channels=in_ch, eps=1e-6)
    self.up = up
    self.down = down
    self.fir = fir
    self.fir_kernel = fir_kernel
    self.Conv_0 = conv3x3(in_ch, out_ch)
    if temb_dim is not None:
      self.Dense_0 = nn.Linear(temb_dim, out_ch)
      self.Dense_0.weight.data = default_init()(self.Dense_0.weight.shape)
      nn.init.zeros_(self.Dense_0.bias)
    self.GroupNorm_1 = nn.GroupNorm(num_groups=min(out_ch // 4, 32), num_channels=out_ch, eps=1e-6)
    self.Dropout_0 = nn.Dropout(dropout)
    self.Conv_1 = conv3x3(out_ch, out_ch, init_scale=init_scale)
    if in_ch != out_ch or up or down:
      self.Conv_2 = conv1x1(in_ch, out_ch)
    self.skip_rescale = skip_rescale
    self.act = act
    self.in_ch = in_ch
    self.out_ch = out_ch

  def emit(self, g, x1, x2, temb, temb_col, name='res'):
    h = g.gn_act(x1, x2, self.GroupNorm_0, act=True, name=name + '.gn0')
    if self.up or self.down:
      assert x2 is None, 'resampling blocks take a single input tensor'
      if self.up:
        r = (lambda t, n: uds.emit_upsample_2d(g, t, self.fir_kernel, factor=2, name=n)) if self.fir \
          else (lambda t, n: uds.emit_naive_upsample_2d(g, t, name=n))
      else:
        r = (lambda t, n: uds.emit_downsample_2d(g, t, self.fir_kernel, factor=2, name=n)) if self.fir \
          else (lambda t, n: uds.emit_naive_downsample_2d(g, t, name=n))
      h = r(h, name + '.rs_h')
      x1 = r(x1, name + '.rs_x')
    h = layers.conv_emit(g, self.Conv_0, h, temb=temb, temb_col=temb_col, name=name + '.conv0')
    h = g.gn_act(h, None, self.GroupNorm_1, act=True, drop_p=self.Dropout_0.p, name=name + '.gn1')
    if self.in_ch != self.out_ch or self.up or self.down:
      xs = layers.conv_emit(g, self.Conv_2, x1, x2, name=name + '.sc')
    else:
      assert x2 is None
      xs = x1
    return layers.conv_emit(g, self.Conv_1, h, res=xs, out_div=SQRT2 if self.skip_rescale else 1.0,
                            name=name + '.conv1')

  def forward(self, x, temb=None):
    _graph_only('ResnetBlockBigGANpp')
