"""Predictor-corrector and probability-flow ODE samplers (reference: sampling.py).

Registries (:30-77), ``get_sampling_fn`` (:80-125), ``Predictor`` / ``Corrector`` ABCs (:128-182),
predictors ``euler_maruyama`` / ``reverse_diffusion`` / ``ancestral_sampling`` / ``none``
(:185-260), correctors ``langevin`` / ``ald`` / ``none`` (:263-340), ``get_pc_sampler`` (:365-433)
and ``get_ode_sampler`` (:436-504) keep the reference's names, signatures and update rules.

Every score evaluation inside the loops is one planned-graph launch sequence of the HIP engine
(models/ncsnpp.py -> engine/executor.py); the per-step updates are a handful of element-wise ops
on the [B,3,H,W] state, all on the device -- the state never leaves HBM during the PC loop.
The loop order is the reference's: corrector first, then predictor (:426-427); the final
denoising step is a noise-free reverse-diffusion step from ``sde.eps`` to 0 (:402-408).

Multi-GPU: sampling shards as independent replicas (no exchange); Langevin's norms are means
over the *local* batch, as they are per DataParallel-gathered batch in the reference (:286-287).
"""
import abc
import functools

import numpy as np
import torch
from scipy import integrate

from . import sde_lib
from .engine import rk45
from .models import utils as mutils
from .models.utils import from_flattened_numpy, get_score_fn, to_flattened_numpy

try:
  from tqdm import tqdm
except ImportError:  # pragma: no cover
  def tqdm(it, **kw):
    return it

_CORRECTORS = {}
_PREDICTORS = {}


def _make_register(table):
  def register(cls=None, *, name=None):
    def _register(cls):
      local_name = cls.__name__ if name is None else name
      if local_name in table:
        raise ValueError(f'Already registered model with name: {local_name}')
      table[local_name] = cls
      return cls
    return _register if cls is None else _register(cls)
  return register


register_predictor = _make_register(_PREDICTORS)
register_corrector = _make_register(_CORRECTORS)


def get_predictor(name):
  return _PREDICTORS[name]


def get_corrector(name):
  return _CORRECTORS[name]


def get_sampling_fn(config, sde, shape, inverse_scaler, eps):
  """``sampling_fn(model) -> (samples, nfe)`` for ``config.sampling.method`` (sampling.py:80-125)."""
  sampler_name = config.sampling.method
  if sampler_name.lower() == 'ode':
    return get_ode_sampler(config=config, sde=sde, shape=shape, inverse_scaler=inverse_scaler,
                           denoise=config.sampling.noise_removal, eps=eps, device=config.device)
  if sampler_name.lower() == 'pc':
    predictor = get_predictor(config.sampling.predictor.lower())
    corrector = get_corrector(config.sampling.corrector.lower())
    return get_pc_sampler(config=config, sde=sde, shape=shape, predictor=predictor, corrector=corrector,
                          inverse_scaler=inverse_scaler, snr=config.sampling.snr,
                          n_steps=config.sampling.n_steps_each,
                          probability_flow=config.sampling.probability_flow,
                          continuous=config.training.continuous,
                          denoise=config.sampling.noise_removal, eps=eps, device=config.device)
  raise ValueError(f"Sampler name {sampler_name} unknown.")


class Predictor(abc.ABC):
  """Abstract predictor; builds the reverse SDE/ODE once (sampling.py:128-157)."""

  def __init__(self, sde, score_fn, probability_flow=False, logsnr_model=None):
    super().__init__()
    self.sde = sde
    if logsnr_model is None:
      lambda_ = 0. if probability_flow else 1.
      self.rsde = sde.reverse(score_fn, probability_flow, lambda_=lambda_)
    else:
      self.rsde = sde.reverse(score_fn, logsnr_model, probability_flow)
    self.score_fn = score_fn

  @abc.abstractmethod
  def update_fn(self, x, t, next_t=None):
    """One predictor update -> (x, x_mean)."""


class Corrector(abc.ABC):
  """Abstract corrector (sampling.py:160-182)."""

  def __init__(self, sde, score_fn, snr, n_steps):
    super().__init__()
    self.sde = sde
    self.score_fn = score_fn
    self.snr = snr
    self.n_steps = n_steps

  @abc.abstractmethod
  def update_fn(self, x, t):
    """One corrector update -> (x, x_mean)."""


@register_predictor(name='euler_maruyama')
class EulerMaruyamaPredictor(Predictor):
  def __init__(self, config, sde, score_fn, probability_flow=False):
    super().__init__(sde, score_fn, probability_flow)

  def update_fn(self, x, t):
    dt = -1. / self.rsde.N
    z = torch.randn_like(x)
    drift, diffusion = self.rsde.sde(x, t)
    x_mean = x + drift * dt
    x = x_mean + diffusion[:, None, None, None] * np.sqrt(-dt) * z
    return x, x_mean


@register_predictor(name='reverse_diffusion')
class ReverseDiffusionPredictor(Predictor):
  def __init__(self, config, sde, score_fn, probability_flow=False, logsnr_model=None):
    super().__init__(sde, score_fn, probability_flow, logsnr_model)
    self.config = config

  def update_fn(self, x, t, next_t=None):
    f, G = self.rsde.discretize(x, t, next_t)
    z = torch.randn_like(x)
    x_mean = x - f
    x = x_mean + G[:, None, None, None] * z
    return x, x_mean


@register_predictor(name='ancestral_sampling')
class AncestralSamplingPredictor(Predictor):
  """Ancestral sampling for VE / VP SDEs (sampling.py:213-249)."""

  def __init__(self, config, sde, score_fn, probability_flow=False):
    super().__init__(sde, score_fn, probability_flow)
    if not isinstance(sde, sde_lib.VPSDE) and not isinstance(sde, sde_lib.VESDE):
      raise NotImplementedError(f"SDE class {sde.__class__.__name__} not yet supported.")
    assert not probability_flow, "Probability flow not supported by ancestral sampling"

  def vesde_update_fn(self, x, t):
    sde = self.sde
    timestep = (t * (sde.N - 1) / sde.T).long()
    sigmas = sde.discrete_sigmas.to(t.device)
    sigma = sigmas[timestep]
    adjacent_sigma = torch.where(timestep == 0, torch.zeros_like(t), sigmas[timestep - 1])
    score = self.score_fn(x, t)
    x_mean = x + score * (sigma ** 2 - adjacent_sigma ** 2)[:, None, None, None]
    std = torch.sqrt((adjacent_sigma ** 2 * (sigma ** 2 - adjacent_sigma ** 2)) / (sigma ** 2))
    noise = torch.randn_like(x)
    x = x_mean + std[:, None, None, None] * noise
    return x, x_mean

  def vpsde_update_fn(self, x, t):
    sde = self.sde
    timestep = (t * (sde.N - 1) / sde.T).long()
    beta = sde.discrete_betas.to(t.device)[timestep]
    score = self.score_fn(x, t)
    x_mean = (x + beta[:, None, None, None] * score) / torch.sqrt(1. - beta)[:, None, None, None]
    noise = torch.randn_like(x)
    x = x_mean + torch.sqrt(beta)[:, None, None, None] * noise
    return x, x_mean

  def update_fn(self, x, t):
    if isinstance(self.sde, sde_lib.VESDE):
      return self.vesde_update_fn(x, t)
    if isinstance(self.sde, sde_lib.VPSDE):
      return self.vpsde_update_fn(x, t)


@register_predictor(name='none')
class NonePredictor(Predictor):
  """Identity predictor."""

  def __init__(self, sde, score_fn, probability_flow=False):
    pass

  def update_fn(self, x, t):
    return x, x


def _vp_alpha(sde, t):
  if isinstance(sde, (sde_lib.VPSDE, sde_lib.subVPSDE)):
    timestep = (t * (sde.N - 1) / sde.T).long()
    return sde.alphas.to(t.device)[timestep]
  return torch.ones_like(t)


def _check_corrector_sde(sde):
  if not isinstance(sde, (sde_lib.VPSDE, sde_lib.VESDE, sde_lib.subVPSDE)):
    raise NotImplementedError(f"SDE class {sde.__class__.__name__} not yet supported.")


@register_corrector(name='langevin')
class LangevinCorrector(Corrector):
  """Langevin corrector with batch-mean norms (sampling.py:263-292)."""

  def __init__(self, sde, score_fn, snr, n_steps):
    super().__init__(sde, score_fn, snr, n_steps)
    _check_corrector_sde(sde)

  def update_fn(self, x, t):
    score_fn = self.score_fn
    target_snr = self.snr
    alpha = _vp_alpha(self.sde, t)
    for i in range(self.n_steps):
      grad = score_fn(x, t)
      noise = torch.randn_like(x)
      grad_norm = torch.norm(grad.reshape(grad.shape[0], -1), dim=-1).mean()
      noise_norm = torch.norm(noise.reshape(noise.shape[0], -1), dim=-1).mean()
      step_size = (target_snr * noise_norm / grad_norm) ** 2 * 2 * alpha
      x_mean = x + step_size[:, None, None, None] * grad
      x = x_mean + torch.sqrt(step_size * 2)[:, None, None, None] * noise
    return x, x_mean


@register_corrector(name='ald')
class AnnealedLangevinDynamics(Corrector):
  """Annealed Langevin dynamics of NCSN/NCSNv2 (sampling.py:295-329)."""

  def __init__(self, sde, score_fn, snr, n_steps):
    super().__init__(sde, score_fn, snr, n_steps)
    _check_corrector_sde(sde)

  def update_fn(self, x, t):
    score_fn = self.score_fn
    target_snr = self.snr
    alpha = _vp_alpha(self.sde, t)
    std = self.sde.marginal_prob(x, t)[1]
    for i in range(self.n_steps):
      grad = score_fn(x, t)
      noise = torch.randn_like(x)
      step_size = (target_snr * std) ** 2 * 2 * alpha
      x_mean = x + step_size[:, None, None, None] * grad
      x = x_mean + noise * torch.sqrt(step_size * 2)[:, None, None, None]
    return x, x_mean


@register_corrector(name='none')
class NoneCorrector(Corrector):
  """Identity corrector."""

  def __init__(self, sde, score_fn, snr, n_steps):
    pass

  def update_fn(self, x, t):
    return x, x


def shared_predictor_update_fn(x, t, sde, model, predictor, probability_flow, continuous, config):
  """Build the predictor for ``model`` and apply one update (sampling.py:343-351)."""
  score_fn = mutils.get_score_fn(config, sde, model, train=False, continuous=continuous)
  if predictor is None:
    predictor_obj = NonePredictor(sde, score_fn, probability_flow)
  else:
    predictor_obj = predictor(config, sde, score_fn, probability_flow)
  return predictor_obj.update_fn(x, t)


def shared_corrector_update_fn(x, t, sde, model, corrector, continuous, snr, n_steps, config):
  """Build the corrector for ``model`` and apply one update (sampling.py:354-362)."""
  score_fn = mutils.get_score_fn(config, sde, model, train=False, continuous=continuous)
  if corrector is None:
    corrector_obj = NoneCorrector(sde, score_fn, snr, n_steps)
  else:
    corrector_obj = corrector(sde, score_fn, snr, n_steps)
  return corrector_obj.update_fn(x, t)


def get_pc_sampler(config, sde, shape, predictor, corrector, inverse_scaler, snr, n_steps=1,
                   probability_flow=False, continuous=False, denoise=True, eps=1e-3, device='cuda'):
  """Predictor-corrector sampler (sampling.py:365-433)."""
  predictor_update_fn = functools.partial(shared_predictor_update_fn, sde=sde, predictor=predictor,
                                          probability_flow=probability_flow, continuous=continuous,
                                          config=config)
  corrector_update_fn = functools.partial(shared_corrector_update_fn, sde=sde, corrector=corrector,
                                          continuous=continuous, snr=snr, n_steps=n_steps, config=config)

  def denoise_update_fn(model, x):
    score_fn = get_score_fn(config, sde, model, train=False, continuous=True)
    predictor_obj = ReverseDiffusionPredictor(config, sde, score_fn, probability_flow=True)
    vec_eps = torch.ones(x.shape[0], device=x.device) * sde.eps
    _, x = predictor_obj.update_fn(x, vec_eps, torch.zeros_like(vec_eps))
    return x

  def pc_sampler(model):
    with torch.no_grad():
      x = sde.prior_sampling(shape).to(device)
      timesteps = torch.linspace(sde.T, eps, sde.N, device=device)
      for i in tqdm(range(sde.N)):
        t = timesteps[i]
        vec_t = torch.ones(shape[0], device=t.device) * t
        x, x_mean = corrector_update_fn(x, vec_t, model=model)
        x, x_mean = predictor_update_fn(x, vec_t, model=model)
      x_mean = x = denoise_update_fn(model, x_mean if denoise else x)
      return inverse_scaler(x_mean if denoise else x), sde.N * (n_steps + 1)

  return pc_sampler


def get_ode_sampler(config, sde, shape, inverse_scaler, denoise=False, rtol=1e-5, atol=1e-5,
                    method='RK45', eps=1e-3, device='cuda'):
  """Probability-flow ODE sampler (sampling.py:436-504): SciPy's RK45 algorithm, float64 solver state.

  `method='RK45'` (every config) runs the solver on device tensors (engine/rk45.py); other methods go through
  scipy.integrate.solve_ivp on the host exactly as the reference does."""

  def denoise_update_fn(model, x):
    score_fn = get_score_fn(config, sde, model, train=False, continuous=True)
    predictor_obj = ReverseDiffusionPredictor(config, sde, score_fn, probability_flow=False)
    vec_eps = torch.ones(x.shape[0], device=x.device) * sde.eps
    _, x = predictor_obj.update_fn(x, vec_eps, torch.zeros_like(vec_eps))
    return x

  def drift_fn(model, x, t):
    score_fn = get_score_fn(config, sde, model, train=False, continuous=True)
    rsde = sde.reverse(score_fn, probability_flow=True, lambda_=0.)
    return rsde.sde(x, t)[0]

  def ode_sampler(model):
    with torch.no_grad():
      x = sde.prior_sampling(shape).to(device)

      if method == 'RK45':
        # SciPy's RK45 restated on device tensors (engine/rk45.py: same tableau, error norm and step controller,
        # checked against SciPy itself): the float64 solver state stays on the device instead of making a host
        # round trip per network evaluation; the network sees the same float32 cast of it as in the reference.
        def ode_func(t, y):
          xt = y.reshape(shape).to(torch.float32)
          vec_t = torch.ones(shape[0], device=xt.device) * t
          return drift_fn(model, xt, vec_t).reshape(-1).to(torch.float64)

        y, nfe = rk45.solve_ivp_rk45(ode_func, (sde.T, eps), x.reshape(-1).to(torch.float64), rtol=rtol, atol=atol)
        x = y.reshape(shape).to(device).type(torch.float32)
      else:
        def ode_func(t, x):
          x = from_flattened_numpy(x, shape).to(device).type(torch.float32)
          vec_t = torch.ones(shape[0], device=x.device) * t
          drift = drift_fn(model, x, vec_t)
          return to_flattened_numpy(drift)

        solution = integrate.solve_ivp(ode_func, (sde.T, eps), to_flattened_numpy(x),
                                       rtol=rtol, atol=atol, method=method)
        nfe = solution.nfev
        x = torch.tensor(solution.y[:, -1]).reshape(shape).to(device).type(torch.float32)
      if denoise:
        x = denoise_update_fn(model, x)
      x = inverse_scaler(x)
      return x, nfe

  return ode_sampler
