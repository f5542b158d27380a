"""Model registry and the score-function wrappers (reference: models/utils.py).

Kept verbatim at the interface level so a ``run_lib.py``-style driver works unchanged:
``register_model`` / ``get_model`` (:25-48), ``get_sigmas`` (:51-62), ``get_ddpm_params`` (:65-86),
``create_model`` (:89-95), ``get_model_fn`` (:97-126), ``get_score_fn`` (:128-190) and the
flatten helpers (:193-200).

MI355X-specific difference, by design: the reference wraps the model in
``torch.nn.DataParallel`` (one process, threads, per-forward broadcast of all 247 MB of
parameters).  Here every GPU runs its own process; :class:`DataParallel` below only keeps the
``module.`` key prefix of the reference's checkpoints and, when ``torch.distributed`` is
initialised, averages gradients with one bucketed RCCL all-reduce per step
(see ``engine/ddp.py``).
"""
import numpy as np
import torch

from .. import sde_lib

_MODELS = {}


def register_model(cls=None, *, name=None):
  """Decorator registering a model class under ``name`` (default: the class name)."""

  def add(model_cls):
    key = name if name is not None else model_cls.__name__
    if key in _MODELS:
      raise ValueError(f'Already registered model with name: {key}')
    _MODELS[key] = model_cls
    return model_cls

  return add if cls is None else add(cls)


def get_model(name):
  return _MODELS[name]


def get_sigmas(config):
  """Geometric ladder of SMLD noise levels, sigma_max -> sigma_min (models/utils.py:51-62)."""
  return np.exp(np.linspace(np.log(config.model.sigma_max), np.log(config.model.sigma_min),
                            config.model.num_scales))


def get_ddpm_params(config):
  """DDPM beta/alpha tables over 1000 steps (models/utils.py:65-86)."""
  steps, m = 1000, config.model
  lo, hi = m.beta_min / m.num_scales, m.beta_max / m.num_scales
  betas = np.linspace(lo, hi, steps, dtype=np.float64)
  alphas = 1. - betas
  cum = np.cumprod(alphas, axis=0)
  return dict(betas=betas, alphas=alphas, alphas_cumprod=cum, sqrt_alphas_cumprod=np.sqrt(cum),
              sqrt_1m_alphas_cumprod=np.sqrt(1. - cum), beta_min=lo * (steps - 1), beta_max=hi * (steps - 1),
              num_diffusion_timesteps=steps)


class DataParallel(torch.nn.Module):
  """Single-device stand-in for ``torch.nn.DataParallel`` (models/utils.py:94).

  state_dict keys keep the ``module.`` prefix of the reference's checkpoints.  Data
  parallelism is one process per GPU: gradient exchange happens in ``losses.optimize_fn`` via
  ``engine.ddp`` when a process group exists, never by replicating the module across threads.
  """

  def __init__(self, module):
    super().__init__()
    self.module = module

  def forward(self, *args, **kwargs):
    return self.module(*args, **kwargs)


def create_model(config, sde):
  """Instantiate ``config.model.name`` on ``config.device`` (models/utils.py:89-95)."""
  score_model = get_model(config.model.name)(config, sde)
  score_model = score_model.to(config.device)
  if hasattr(score_model, 'engine'):
    # bind the parameters to the flat HBM buffers now, so the optimizer and the EMA created right after
    # (utils.load_model) see the final storage.  Loads libstk.so: raises if the HIP library is missing.
    score_model.engine().ensure_flat()
  return DataParallel(score_model)


def frozen_weights(model):
  """Context manager for loops that evaluate `model` many times on fixed parameters (samplers, likelihood ODEs): the
  engine prepares the convolution weights once instead of once per evaluation.  A no-op for models without an engine."""
  import contextlib
  inner = getattr(model, 'module', model)
  engine = getattr(inner, 'engine', None)
  return engine().frozen_weights() if callable(engine) else contextlib.nullcontext()


def get_model_fn(model, train=False):
  """Callable running the model in train or eval mode, re-asserted on every call (models/utils.py:97-126)."""

  def model_fn(x, labels):
    model.train() if train else model.eval()
    return model(x, labels)

  return model_fn


def _vp_time_labels(config, sde, t):
  """Conditioning of a continuously-trained VP network: 999 t, or the unbounded parametrisation that rescales the
  antiderivative of g^2/sigma^2 onto [0, 999] (models/utils.py:149-154)."""
  tr = config.training
  if not tr.unbounded_parametrization:
    return t * 999
  sc = tr.stabilizing_constant
  a0 = sde.antiderivative(1e-5, stabilizing_constant=sc)
  return (sde.antiderivative(t, stabilizing_constant=sc) - a0) / (sde.antiderivative(sde.T, stabilizing_constant=sc) - a0) * 999.


def _vp_score_fn(config, sde, model_fn, continuous):
  sub = isinstance(sde, sde_lib.subVPSDE)

  def score_fn(x, t, logsnr_model=None, logsnr=None):
    if continuous or sub:
      labels = _vp_time_labels(config, sde, t)
      std = sde.marginal_prob(torch.zeros_like(x), t)[1]
      out = model_fn(x, labels)
    else:                                           # discrete DDPM ladder
      labels = t * (sde.N - 1)
      out = model_fn(x, labels)
      std = sde.sqrt_1m_alphas_cumprod.to(labels.device)[labels.long()]
    return - out / std[:, None, None, None] if config.training.ddpm_score else out

  return score_fn


def _ve_score_fn(sde, model_fn, continuous):
  def score_fn(x, t):
    if continuous:
      labels = sde.marginal_prob(torch.zeros_like(x), t)[1]          # sigma(t)
    else:
      labels = sde.T - t
      labels *= sde.N - 1
      labels = torch.round(labels).long()
    return model_fn(x, labels)

  return score_fn


def get_raw_fn(config, sde, model, train=False, continuous=False):
  """The pieces of :func:`get_score_fn` for callers that fuse what follows the network (losses.py's fused loss path):
  ``raw_fn(x, t) -> network output`` with exactly the conditioning labels `get_score_fn` computes, and ``neg_over_std``:
  whether the score is ``-output / std(t)`` (VP family with ``training.ddpm_score``) or the output itself.  Returns
  ``(None, None)`` for the combinations it does not cover (discrete ladders)."""
  model_fn = get_model_fn(model, train=train)
  if isinstance(sde, (sde_lib.VPSDE, sde_lib.subVPSDE)):
    if not (continuous or isinstance(sde, sde_lib.subVPSDE)):
      return None, None
    return (lambda x, t: model_fn(x, _vp_time_labels(config, sde, t))), bool(config.training.ddpm_score)
  if isinstance(sde, (sde_lib.VESDE, sde_lib.reciprocal_VESDE)) and continuous:
    tiny = lambda x: torch.zeros((x.shape[0], 1, 1, 1), dtype=x.dtype, device=x.device)
    return (lambda x, t: model_fn(x, sde.marginal_prob(tiny(x), t)[1])), False      # labels = sigma(t)
  return None, None


def get_score_fn(config, sde, model, train=False, continuous=False):
  """Turn the raw network into a score function s(x, t) (models/utils.py:128-190).

  VP / subVP: labels = 999 t (continuous), score = -net/std when ``training.ddpm_score``;
  VE / RVE:   labels = sigma(t) (continuous), the network output is already the score."""
  model_fn = get_model_fn(model, train=train)
  if isinstance(sde, (sde_lib.VPSDE, sde_lib.subVPSDE)):
    return _vp_score_fn(config, sde, model_fn, continuous)
  if isinstance(sde, (sde_lib.VESDE, sde_lib.reciprocal_VESDE)):
    return _ve_score_fn(sde, model_fn, continuous)
  raise NotImplementedError(f"SDE class {sde.__class__.__name__} not yet supported.")


def to_flattened_numpy(x):
  return x.detach().cpu().numpy().reshape((-1,))


def from_flattened_numpy(x, shape):
  return torch.from_numpy(x.reshape(shape))
