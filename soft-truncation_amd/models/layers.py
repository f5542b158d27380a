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
import torch.nn.functional as F


def get_act(config):
  """Activation named by ``config.model.nonlinearity``.  The engine implements 'swish' (SiLU),
  the only value used by the NCSN++ configs; the others are returned for API parity."""
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


def variance_scaling(scale, mode, distribution, in_axis=1, out_axis=0, dtype=torch.float32, device='cpu'):
  """JAX-style variance-scaling initializer (models/layers.py:54-85)."""

  def _compute_fans(shape):
    receptive_field_size = np.prod(shape) / shape[in_axis] / shape[out_axis]
    return shape[in_axis] * receptive_field_size, shape[out_axis] * receptive_field_size

  def init(shape, dtype=dtype, device=device):
    fan_in, fan_out = _compute_fans(shape)
    if mode == "fan_in":
      denominator = fan_in
    elif mode == "fan_out":
      denominator = fan_out
    elif mode == "fan_avg":
      denominator = (fan_in + fan_out) / 2
    else:
      raise ValueError("invalid mode for variance scaling initializer: {}".format(mode))
    variance = scale / denominator
    if distribution == "normal":
      return torch.randn(*shape, dtype=dtype, device=device) * np.sqrt(variance)
    if distribution == "uniform":
      return (torch.rand(*shape, dtype=dtype, device=device) * 2. - 1.) * np.sqrt(3 * variance)
    raise ValueError("invalid distribution for variance scaling initializer")

  return init


def default_init(scale=1.):
  """DDPM initialisation: fan-avg uniform; ``scale == 0`` means 1e-10 (models/layers.py:88-91)."""
  scale = 1e-10 if scale == 0 else scale
  return variance_scaling(scale, 'fan_avg', 'uniform')


def ddpm_conv1x1(in_planes, out_planes, stride=1, bias=True, init_scale=1., padding=0):
  conv = nn.Conv2d(in_planes, out_planes, kernel_size=1, stride=stride, padding=padding, bias=bias)
  conv.weight.data = default_init(init_scale)(conv.weight.data.shape)
  nn.init.zeros_(conv.bias)
  return conv


def ddpm_conv3x3(in_planes, out_planes, stride=1, bias=True, dilation=1, init_scale=1., padding=1):
  conv = nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=padding,
                   dilation=dilation, bias=bias)
  conv.weight.data = default_init(init_scale)(conv.weight.data.shape)
  nn.init.zeros_(conv.bias)
  return conv


def get_timestep_embedding(timesteps, embedding_dim, max_positions=10000):
  """Host-side statement of the sinusoidal embedding (models/layers.py:515-529).  Inside the
  network the same values are produced by ``stk_timestep_embedding_f32``."""
  assert len(timesteps.shape) == 1
  half_dim = embedding_dim // 2
  emb = math.log(max_positions) / (half_dim - 1)
  emb = torch.exp(torch.arange(half_dim, dtype=torch.float32, device=timesteps.device) * -emb)
  emb = timesteps.float()[:, None] * emb[None, :]
  emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=1)
  if embedding_dim % 2 == 1:
    emb = F.pad(emb, (0, 1), mode='constant')
  assert emb.shape == (timesteps.shape[0], embedding_dim)
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
