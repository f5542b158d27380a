"""Configuration trees for the five BASELINE.json workloads.

The reference describes a run with an ``ml_collections.ConfigDict`` built by a
``get_config()`` function per file under ``configs/``.  ``ml_collections`` is
not a dependency of this package: every consumer here only needs attribute
access (``config.model.nf``), so :class:`ConfigDict` below is a tiny
attribute-dict and a real ``ml_collections.ConfigDict`` works just as well.

Only *values* are restated here, keyed by the reference
file they come from:

* ``cifar10_ddpmpp_nll_st``  <- configs/default_cifar10_configs.py:5-100 +
  configs/vp/CIFAR10/ddpmpp_nll_st.py:22-70          (BASELINE configs[0], [1])
* ``celeba_uncsnpp_st``      <- configs/default_celeba_configs.py +
  configs/ve/CELEBA/uncsnpp_st.py                     (BASELINE configs[2])
* ``imagenet32_ddpmpp_st``   <- configs/vp/IMAGENET32/ddpmpp_st.py (configs[3])
* ``celebahq_uncsnpp_st``    <- configs/default_lsun_configs.py +
  configs/ve/celebahq/uncsnpp_st.py                   (BASELINE configs[4])
"""
import copy

import torch


class ConfigDict(dict):
  """Attribute-style nested dict (stand-in for ml_collections.ConfigDict)."""

  def __getattr__(self, name):
    try:
      return self[name]
    except KeyError as e:
      raise AttributeError(name) from e

  def __setattr__(self, name, value):
    self[name] = value

  def __deepcopy__(self, memo):
    out = ConfigDict()
    for k, v in self.items():
      out[k] = copy.deepcopy(v, memo)
    return out

  def to_dict(self):
    return {k: (v.to_dict() if isinstance(v, ConfigDict) else v) for k, v in self.items()}


def _tree(d):
  out = ConfigDict()
  for k, v in d.items():
    out[k] = _tree(v) if isinstance(v, dict) else v
  return out


def _default_device():
  return torch.device('cuda:0') if torch.cuda.is_available() else torch.device('cpu')


# ---------------------------------------------------------------------------
# shared defaults (configs/default_cifar10_configs.py:5-100)
# ---------------------------------------------------------------------------
def _cifar10_defaults():
  return _tree(dict(
    training=dict(
      batch_size=128, n_iters=13000001, snapshot_freq=100000, log_freq=100, eval_freq=100,
      snapshot_freq_for_preemption=10000, snapshot_sampling=False,
      likelihood_weighting=True, continuous=True, reduce_mean=False,
      importance_sampling=True, unbounded_parametrization=False, ddpm_score=True,
      st=False, truncation_time=1e-5, num_train_data=50000, reconstruction_loss=False,
      stabilizing_constant=1e-3, whatever_sampling=False, mixed=False, ddpm_weight=0.01,
      balanced=False),
    sampling=dict(
      n_steps_each=1, noise_removal=True, probability_flow=False, snr=0.16,
      batch_size=1024, truncation_time=1e-5, sample_more=True),
    eval=dict(
      begin_ckpt=9, end_ckpt=26, batch_size=200, enable_sampling=False, num_samples=50000,
      enable_loss=True, enable_bpd=False, bpd_dataset='test', num_test_data=10000,
      residual=True, lambda_=0.0, probability_flow=True, nelbo_iter=0, nll_iter=0),
    data=dict(
      dataset='CIFAR10', image_size=32, random_flip=True, centered=False,
      dequantization='none', num_channels=3),
    model=dict(
      sigma_min=0.01, sigma_max=50, num_scales=1000, beta_min=0.1, beta_max=20.,
      dropout=0.1, embedding_type='fourier', auxiliary_resblock=True, attention=True,
      fourier_feature=False, lsgm=False),
    optim=dict(
      weight_decay=0.0, optimizer='Adam', lr=2e-4, beta1=0.9, eps=1e-8, warmup=5000,
      grad_clip=1., num_micro_batch=1, amsgrad=False),
    seed=42,
  ))


def _ncsnpp_common(model):
  model.name = 'ncsnpp'
  model.normalization = 'GroupNorm'
  model.nonlinearity = 'swish'
  model.nf = 128
  model.ch_mult = (1, 2, 2, 2)
  model.num_res_blocks = 4
  model.attn_resolutions = (16,)
  model.resamp_with_conv = True
  model.conditional = True
  model.fir_kernel = [1, 3, 3, 1]
  model.skip_rescale = True
  model.resblock_type = 'biggan'
  model.progressive_combine = 'sum'
  model.attention_type = 'ddpm'
  model.init_scale = 0.
  model.fourier_scale = 16
  model.conv_size = 3


def cifar10_ddpmpp_nll_st():
  """DDPM++ (VP) CIFAR-10 32x32 -- configs/vp/CIFAR10/ddpmpp_nll_st.py:22-70."""
  config = _cifar10_defaults()
  t = config.training
  t.sde = 'vpsde'
  t.continuous = True
  t.reduce_mean = True
  t.st = True
  t.k = 1.0
  t.likelihood_weighting = False
  t.truncation_time = 1e-5
  s = config.sampling
  s.method = 'ode'
  s.predictor = 'euler_maruyama'
  s.corrector = 'none'
  config.data.centered = True
  m = config.model
  _ncsnpp_common(m)
  m.scale_by_sigma = False
  m.ema_rate = 0.9999
  m.fir = False
  m.progressive = 'none'
  m.progressive_input = 'none'
  m.embedding_type = 'positional'
  config.device = _default_device()
  return config


def imagenet32_ddpmpp_st():
  """DDPM++ (VP) ImageNet32 -- configs/vp/IMAGENET32/ddpmpp_st.py (dropout 0)."""
  config = cifar10_ddpmpp_nll_st()
  config.training.num_train_data = 1281149
  config.data.dataset = 'IMAGENET32'
  config.model.dropout = 0.
  config.eval.num_test_data = 49999
  return config


def celeba_uncsnpp_st():
  """UNCSN++ (RVE) CelebA 64x64 -- configs/default_celeba_configs.py + ve/CELEBA/uncsnpp_st.py."""
  config = _cifar10_defaults()
  t = config.training
  t.n_iters = 1300001
  t.snapshot_freq = 50000
  t.log_freq = 50
  t.snapshot_sampling = True
  t.likelihood_weighting = False
  t.num_train_data = 162770
  t.sde = 'reciprocal_vesde'
  t.continuous = True
  t.importance_sampling = False
  t.st = True
  t.truncation_time = 1e-5
  t.model_mode = 'reciprocal'
  t.eta = 1e-3
  s = config.sampling
  s.snr = 0.17
  s.batch_size = 512
  s.method = 'pc'
  s.predictor = 'reverse_diffusion'
  s.corrector = 'langevin'
  e = config.eval
  e.begin_ckpt = 1
  e.batch_size = 1024
  e.num_test_data = 19962
  d = config.data
  d.dataset = 'CELEBA'
  d.image_size = 64
  d.centered = False
  m = config.model
  m.sigma_max = 90.
  _ncsnpp_common(m)
  m.scale_by_sigma = True
  m.sigma_begin = 90
  m.ema_rate = 0.999
  m.fir = True
  m.progressive = 'none'
  m.progressive_input = 'residual'
  m.fourier_feature = False
  m.sigma_min = 1e-3
  config.device = _default_device()
  return config


def celebahq_uncsnpp_st():
  """NCSN++ (VE) CelebA-HQ 256x256 -- configs/default_lsun_configs.py + ve/celebahq/uncsnpp_st.py."""
  config = _cifar10_defaults()
  t = config.training
  t.batch_size = 64
  t.n_iters = 24000001
  t.snapshot_freq = 200000
  t.log_freq = 1000
  t.eval_freq = 500
  t.snapshot_freq_for_preemption = 5000
  t.likelihood_weighting = False
  t.importance_sampling = False
  t.num_train_data = 162770
  t.sde = 'vesde'
  t.continuous = True
  t.st = True
  t.k = 2.0
  t.truncation_time = 1e-5
  s = config.sampling
  s.snr = 0.075
  s.batch_size = 16
  s.truncation_time = 1e-3
  del s['sample_more']
  s.method = 'pc'
  s.predictor = 'reverse_diffusion'
  s.corrector = 'langevin'
  s.probability_flow = False
  e = config.eval
  e.begin_ckpt = 50
  e.end_ckpt = 96
  e.batch_size = 512
  e.enable_sampling = True
  for k in ('num_test_data', 'residual', 'lambda_', 'probability_flow', 'nelbo_iter', 'nll_iter'):
    del e[k]
  d = config.data
  d.dataset = 'CelebAHQ'
  d.image_size = 256
  d.centered = False
  m = config.model
  m.num_scales = 2000
  m.dropout = 0.
  _ncsnpp_common(m)
  m.sigma_max = 348
  m.scale_by_sigma = True
  m.ema_rate = 0.999
  m.ch_mult = (1, 1, 2, 2, 2, 2, 2)
  m.num_res_blocks = 2
  m.fir = True
  m.progressive = 'output_skip'
  m.progressive_input = 'input_skip'
  m.fourier_feature = False
  config.device = _default_device()
  return config


BASELINE_CONFIGS = {
  'cifar10_ddpmpp_nll_st': cifar10_ddpmpp_nll_st,
  'imagenet32_ddpmpp_st': imagenet32_ddpmpp_st,
  'celeba_uncsnpp_st': celeba_uncsnpp_st,
  'celebahq_uncsnpp_st': celebahq_uncsnpp_st,
}


def get_config(name):
  return BASELINE_CONFIGS[name]()


def tiny(config, nf=16, ch_mult=(1, 2), num_res_blocks=1, image_size=16, attn_resolutions=(8,),
         dropout=0.0):
  """Shrink a BASELINE config to fixture size (same flag combination, small tensors)."""
  config = copy.deepcopy(config)
  config.model.nf = nf
  config.model.ch_mult = tuple(ch_mult)
  config.model.num_res_blocks = num_res_blocks
  config.model.attn_resolutions = tuple(attn_resolutions)
  config.model.dropout = dropout
  config.data.image_size = image_size
  return config
