"""Shared helpers for the model-level tests (CPU host-logic tests with the checker backend, and the
GPU parity tests with the HIP backend)."""
import contextlib
import copy

import numpy as np
import torch

import ref_torch


def tiny_config(st, family, dropout=0.0, device='cpu'):
  """Fixture-sized versions of the three live flag combinations."""
  if family == 'vp':        # DDPM++: positional embedding, naive resampling, VP SDE, IS loss
    cfg = st.configs.tiny(st.configs.cifar10_ddpmpp_nll_st(), dropout=dropout)
  elif family == 'rve':     # UNCSN++: fourier, FIR, progressive_input residual, RVE SDE
    cfg = st.configs.tiny(st.configs.celeba_uncsnpp_st(), dropout=dropout)
  elif family == 've':      # NCSN++ 256-style: fourier, FIR, input_skip / output_skip, VE SDE
    cfg = st.configs.tiny(st.configs.celebahq_uncsnpp_st(), ch_mult=(1, 1, 2), dropout=dropout)
  elif family == 'wide':    # DDPM++ with enough channels (96 / 192) for the split (fp16 two-way) conv kernels, their K-split and
    # few-tile variants and the prepared-weight path; the other families stay on the f32-input kernels
    cfg = st.configs.tiny(st.configs.cifar10_ddpmpp_nll_st(), nf=96, ch_mult=(1, 2), num_res_blocks=1, image_size=16,
                          attn_resolutions=(8,), dropout=dropout)
  elif family in ('vp_elu', 'vp_relu', 'vp_lrelu'):   # the other activations of layers.get_act (models/layers.py:29-41)
    cfg = st.configs.tiny(st.configs.cifar10_ddpmpp_nll_st(), dropout=dropout)
    cfg.model.nonlinearity = family[3:]
  elif family == 've_cat':  # NCSN++ pyramids with Combine(method='cat') (models/layerspp.py:57-72, models/ncsnpp.py:183-184)
    cfg = st.configs.tiny(st.configs.celebahq_uncsnpp_st(), ch_mult=(1, 1, 2), dropout=dropout)
    cfg.model.progressive_combine = 'cat'
  elif family == 'vp_ff':   # model.fourier_feature: FixedFouriereProjection in front of the stem (models/ncsnpp.py:104,156,305)
    cfg = st.configs.tiny(st.configs.cifar10_ddpmpp_nll_st(), dropout=dropout)
    cfg.model.fourier_feature = True
  else:
    raise ValueError(family)
  cfg.device = torch.device(device)
  return cfg


def randomize_(model, seed=0, scale=0.1):
  """Untouched models output ~0 (init_scale=0 -> 1e-10 last convs) and pin nothing: randomise."""
  g = torch.Generator().manual_seed(seed)
  with torch.no_grad():
    for p in model.parameters():
      if p.requires_grad:
        p.copy_((torch.randn(p.shape, generator=g) * scale).to(p.device))


def build_pair(st, cfg, backend, seed=0):
  """Product model (engine on `backend`) and the oracle RefNet with identical weights."""
  dev = torch.device('cuda:0') if backend.is_device else torch.device('cpu')
  cfg = copy.deepcopy(cfg)
  cfg.device = dev
  sde = st.sde_lib.get_sde(cfg, None)
  torch.manual_seed(seed)
  net = st.models.ncsnpp.NCSNpp(cfg, sde)
  net.set_backend(backend)
  net = net.to(dev)
  randomize_(net, seed)
  model = st.models.utils.DataParallel(net)
  net.engine().ensure_flat()
  cfg_cpu = copy.deepcopy(cfg)
  cfg_cpu.device = torch.device('cpu')
  sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
  ref = st.models.utils.DataParallel(ref_torch.RefNet(cfg_cpu, sd))
  return cfg, cfg_cpu, sde, model, ref


class _Draws:
  """Deterministic CPU-generated noise served to whatever device asks for it."""

  def __init__(self, seed):
    self.g = torch.Generator().manual_seed(seed)

  def rand(self, *size, device=None, **kw):
    if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
      size = tuple(size[0])
    return torch.empty(*size).uniform_(generator=self.g).to(device or 'cpu')

  def randn(self, *size, device=None, **kw):
    if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
      size = tuple(size[0])
    return torch.empty(*size).normal_(generator=self.g).to(device or 'cpu')

  def randn_like(self, x, **kw):
    return torch.empty(x.shape).normal_(generator=self.g).to(x.device)

  def randint_like(self, x, low=0, high=None, **kw):
    if high is None:
      low, high = 0, low
    return torch.randint(low, high, tuple(x.shape), generator=self.g).to(device=x.device, dtype=x.dtype)


@contextlib.contextmanager
def patched_rng(seed):
  """Serve torch.rand / randn / randn_like from one CPU stream (CPU and GPU generators differ, so
  parity tests inject the noise explicitly -- SURVEY.md 8(c))."""
  d = _Draws(seed)
  saved = (torch.rand, torch.randn, torch.randn_like, torch.randint_like)
  torch.rand, torch.randn, torch.randn_like, torch.randint_like = d.rand, d.randn, d.randn_like, d.randint_like
  try:
    yield
  finally:
    torch.rand, torch.randn, torch.randn_like, torch.randint_like = saved


def make_state(st, cfg, model):
  opt = st.losses.get_optimizer(cfg, model.parameters())
  ema = st.models.ema.ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate)
  return dict(optimizer=opt, model=model, ema=ema, step=0)


def rel_err(a, b):
  a, b = a.detach().cpu().double(), b.detach().cpu().double()
  return (a - b).abs().max().item() / (b.abs().max().item() + 1e-30)


def grads_close(model, ref, tol):
  """Parameter gradients relative to the largest gradient entry of the whole model (some gradients are
  identically zero in exact arithmetic -- e.g. the key bias of attention -- so per-tensor relative error
  is meaningless for them)."""
  ref_g = {k: p.grad for k, p in ref.named_parameters() if p.grad is not None}
  scale = max(g.abs().max().item() for g in ref_g.values())
  worst, wk = 0.0, None
  names = dict(model.named_parameters())
  for k, p in names.items():
    if not p.requires_grad:
      continue
    rk = k.replace('.', '__').replace('module__', 'module.', 1)
    g = ref_g[rk]
    per = max(g.abs().max().item(), 1e-3 * scale)
    e = (p.grad.detach().cpu() - g).abs().max().item() / per
    if e > worst:
      worst, wk = e, k
  assert worst <= tol, f'parameter gradient mismatch {worst:.3e} at {wk}'
  return worst
