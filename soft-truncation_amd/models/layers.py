"""Shared bricks of the NCSN++ family (reference: models/layers.py, live symbols only).

``get_act`` (:29-41), ``variance_scaling`` / ``default_init`` (:54-91), ``ddpm_conv1x1`` (:100-105),
``ddpm_conv3x3`` (:118-124), ``get_timestep_embedding`` (:515-529), ``NIN`` (:546-555).

The modules below own parameters under the reference's names (so released checkpoints load)
and *emit* their computation into an ``engine.graph.Graph``; the arithmetic itself runs in the
HIP kernels.  The legacy NCSNv1/v2/DDPM blocks of the reference file (:133-507, :558-662) are out
of scope: they are only reachable from model classes that ``create_model`` cannot construct
(SURVEY.md section 0).
"""
import math

import numpy as np
import torch
import torch.nn as nn


def get_act(config):
  """Activation named by ``config.model.nonlinearity`` (models/layers.py:29-41).  The engine runs all four through the
  `act` code of include/stk.h (``act_code``): 'swish' is what every shipped config uses."""
  name = config.model.nonlinearity.lower()
  if name == 'elu':
    return nn.ELU()
  if name == 'relu':
    return nn.ReLU()
  if name == 'lrelu':
    return nn.LeakyReLU(negative_slope=0.2)
  if name == 'swish':
    return nn.SiLU()
  raise NotImplementedError('activation function does not exist!')


ACT_CODES = {'swish': 1, 'relu': 2, 'lrelu': 3, 'elu': 4}     # STK_ACT_* of include/stk.h


def act_code(config):
  name = config.model.nonlinearity.lower()
  if name not in ACT_CODES:
    raise NotImplementedError('activation function does not exist!')
  return ACT_CODES[name]


# fan used as the denominator of the variance, per `mode` (models/layers.py:68-76)
_FAN = {'fan_in': lambda fi, fo: fi, 'fan_out': lambda fi, fo: fo, 'fan_avg': lambda fi, fo: (fi + fo) / 2}


def _fans(shape, in_axis, out_axis):
  """(fan_in, fan_out) of a weight of `shape`: the axis length times the receptive field (every other axis)."""
  field = np.prod(shape) / shape[in_axis] / shape[out_axis]
  return shape[in_axis] * field, shape[out_axis] * field


def scaled_draw_(out, variance, distribution):
  """Fill `out` with zero-mean draws of the given variance from torch's global generator.  The draws and the fp32
  arithmetic are those of the reference's initializer (models/layers.py:77-83: `randn * sqrt(var)`,
  `(rand * 2 - 1) * sqrt(3 var)`), so a model built under the same seed has bit-identical weights
  (tests/test_ref_live.py)."""
  if distribution == 'normal':
    return out.normal_().mul_(float(np.sqrt(variance)))
  if distribution == 'uniform':
    return out.uniform_().mul_(2.).sub_(1.).mul_(float(np.sqrt(3 * variance)))
  raise ValueError("invalid distribution for variance scaling initializer")


def variance_scaling(scale, mode, distribution, in_axis=1, out_axis=0, dtype=torch.float32, device='cpu'):
  """``init(shape) -> tensor`` with variance ``scale / fan`` (the reference's JAX-style initializer factory,
  models/layers.py:54-85; same signature, same error messages for unknown modes / distributions)."""
  if mode not in _FAN:
    pick_fan = None
  else:
    pick_fan = _FAN[mode]

  def init(shape, dtype=dtype, device=device):
    if pick_fan is None:
      raise ValueError("invalid mode for variance scaling initializer: {}".format(mode))
    fan = pick_fan(*_fans(shape, in_axis, out_axis))
    return scaled_draw_(torch.empty(*shape, dtype=dtype, device=device), scale / fan, distribution)

  return init


def default_init(scale=1.):
  """DDPM initialisation: fan-avg uniform; ``scale == 0`` means 1e-10 (models/layers.py:88-91)."""
  return variance_scaling(scale if scale != 0 else 1e-10, 'fan_avg', 'uniform')


def _ddpm_conv(kernel, in_planes, out_planes, init_scale, **conv_kw):
  # nn.Conv2d's own initialisation runs first and consumes the generator exactly as in the reference; then the DDPM one
  conv = nn.Conv2d(in_planes, out_planes, kernel_size=kernel, **conv_kw)
  conv.weight.data = default_init(init_scale)(conv.weight.shape)
  if conv.bias is not None:
    conv.bias.data.zero_()
  return conv


def ddpm_conv1x1(in_planes, out_planes, stride=1, bias=True, init_scale=1., padding=0):
  """1x1 convolution with DDPM initialisation (models/layers.py:100-105)."""
  return _ddpm_conv(1, in_planes, out_planes, init_scale, stride=stride, padding=padding, bias=bias)


def ddpm_conv3x3(in_planes, out_planes, stride=1, bias=True, dilation=1, init_scale=1., padding=1):
  """3x3 convolution with DDPM initialisation (models/layers.py:118-124)."""
  return _ddpm_conv(3, in_planes, out_planes, init_scale, stride=stride, padding=padding, dilation=dilation, bias=bias)


def positional_frequencies(half_dim, max_positions=10000, device=None):
  """f_j = exp(-j ln(max_positions) / (half_dim - 1)), evaluated in fp32 by torch.exp on an fp32 ramp -- the table the
  reference builds on every call (models/layers.py:519-521) and the one ``engine.graph.TimestepEmbedding`` uploads."""
  step = math.log(max_positions) / (half_dim - 1)
  return torch.exp(torch.arange(half_dim, dtype=torch.float32, device=device) * -step)


def get_timestep_embedding(timesteps, embedding_dim, max_positions=10000):
  """Host-side statement of the sinusoidal embedding [sin(t f), cos(t f)] (models/layers.py:515-529).  Inside the
  network the same values are produced by ``stk_timestep_embedding_f32``."""
  assert timesteps.dim() == 1
  half = embedding_dim // 2
  angle = timesteps.float()[:, None] * positional_frequencies(half, max_positions, timesteps.device)[None, :]
  emb = angle.new_zeros(timesteps.shape[0], embedding_dim)      # an odd embedding_dim keeps a zero last column
  emb[:, :half] = torch.sin(angle)
  emb[:, half:2 * half] = torch.cos(angle)
  return emb


def conv_emit(g, conv, x1, x2=None, **kw):
  """Emit an nn.Conv2d (square kernel, symmetric padding) into the graph."""
  return g.conv(x1, x2, conv.weight, conv.bias, w_layout=0, stride=conv.stride[0], pad=conv.padding[0], **kw)


class NIN(nn.Module):
  """1x1 'network in network' layer with an [in, out] weight (models/layers.py:546-555)."""

  def __init__(self, in_dim, num_units, init_scale=0.1):
    super().__init__()
    self.W = nn.Parameter(default_init(scale=init_scale)((in_dim, num_units)), requires_grad=True)
    self.b = nn.Parameter(torch.zeros(num_units), requires_grad=True)

  def emit(self, g, x1, x2=None, **kw):
    return g.conv(x1, x2, self.W, self.b, w_layout=1, stride=1, pad=0, **kw)
