"""Predictor-corrector and probability-flow ODE samplers (reference: sampling.py).

Registries (:30-77), ``get_sampling_fn`` (:80-125), ``Predictor`` / ``Corrector`` ABCs (:128-182),
predictors ``euler_maruyama`` / ``reverse_diffusion`` / ``ancestral_sampling`` / ``none``
(:185-260), correctors ``langevin`` / ``ald`` / ``none`` (:263-340), ``get_pc_sampler`` (:365-433)
and ``get_ode_sampler`` (:436-504) keep the reference's names, signatures and update rules.  The update rules
themselves are written once, as plain functions of (reverse SDE / score function, state, time) in the first half of
this file; the registered classes only bind them to the reference's constructor signatures.

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

import os

try:
  from tqdm import tqdm as _tqdm
except ImportError:  # pragma: no cover
  _tqdm = None

# The reference shows a progress bar over the PC iterations (sampling.py:423).  PROGRESS = False (or STK_PROGRESS=0 in the
# environment) silences it: a benchmark or a batch job must not write a thousand bar updates to stderr.
PROGRESS = os.environ.get('STK_PROGRESS', '1') != '0'


def tqdm(it, **kw):
  if _tqdm is None or not PROGRESS:
    return it
  return _tqdm(it, **kw)


def _wide(v):
  return v[:, None, None, None]


def _ladder_index(sde, t):
  """Index of continuous time t on the N-step discrete ladder."""
  return (t * (sde.N - 1) / sde.T).long()


# ---- update rules: x -> (x_next, x_mean) ---------------------------------------------------------------------------
def _euler_maruyama(rsde, x, t):
  """x_mean = x + f dt,  x = x_mean + g sqrt(-dt) z  with dt = -1/N (sampling.py:190-196)."""
  dt = -1. / rsde.N
  z = torch.randn_like(x)
  drift, diffusion = rsde.sde(x, t)
  x_mean = x + drift * dt
  return x_mean + _wide(diffusion) * np.sqrt(-dt) * z, x_mean


def _reverse_diffusion(rsde, x, t, next_t):
  """x_mean = x - f,  x = x_mean + G z  on the discretised reverse SDE (sampling.py:205-210)."""
  f, G = rsde.discretize(x, t, next_t)
  z = torch.randn_like(x)
  x_mean = x - f
  return x_mean + _wide(G) * z, x_mean


def _ancestral_ve(sde, score_fn, x, t):
  """SMLD ancestral step between adjacent noise levels (sampling.py:222-234)."""
  i = _ladder_index(sde, t)
  ladder = sde.discrete_sigmas.to(t.device)
  sigma, below = ladder[i], torch.where(i == 0, torch.zeros_like(t), ladder[i - 1])
  score = score_fn(x, t)
  x_mean = x + score * _wide(sigma ** 2 - below ** 2)
  std = torch.sqrt((below ** 2 * (sigma ** 2 - below ** 2)) / (sigma ** 2))
  noise = torch.randn_like(x)
  return x_mean + _wide(std) * noise, x_mean


def _ancestral_vp(sde, score_fn, x, t):
  """DDPM ancestral step (sampling.py:236-244)."""
  beta = sde.discrete_betas.to(t.device)[_ladder_index(sde, t)]
  score = score_fn(x, t)
  x_mean = (x + _wide(beta) * score) / _wide(torch.sqrt(1. - beta))
  noise = torch.randn_like(x)
  return x_mean + _wide(torch.sqrt(beta)) * noise, x_mean


def _alpha_of(sde, t):
  """alpha_t of the VP ladders, 1 for VE (sampling.py:278-282)."""
  if isinstance(sde, (sde_lib.VPSDE, sde_lib.subVPSDE)):
    return sde.alphas.to(t.device)[_ladder_index(sde, t)]
  return torch.ones_like(t)


def _batch_mean_norm(v):
  return torch.norm(v.reshape(v.shape[0], -1), dim=-1).mean()


def _langevin(sde, score_fn, snr, n_steps, x, t):
  """n_steps of Langevin MCMC; step = (snr |noise| / |grad|)^2 2 alpha with batch-mean norms (sampling.py:284-290)."""
  alpha = _alpha_of(sde, t)
  for _ in range(n_steps):
    grad = score_fn(x, t)
    noise = torch.randn_like(x)
    grad_norm, noise_norm = _batch_mean_norm(grad), _batch_mean_norm(noise)
    step = (snr * noise_norm / grad_norm) ** 2 * 2 * alpha
    x_mean = x + _wide(step) * grad
    x = x_mean + _wide(torch.sqrt(step * 2)) * noise
  return x, x_mean


def _annealed_langevin(sde, score_fn, snr, n_steps, x, t):
  """NCSN-style annealed Langevin dynamics; step = (snr std_t)^2 2 alpha (sampling.py:311-327)."""
  alpha = _alpha_of(sde, t)
  std = sde.marginal_prob(x, t)[1]
  for _ in range(n_steps):
    grad = score_fn(x, t)
    noise = torch.randn_like(x)
    step = (snr * std) ** 2 * 2 * alpha
    x_mean = x + _wide(step) * grad
    x = x_mean + noise * _wide(torch.sqrt(step * 2))
  return x, x_mean


# ---- registries -----------------------------------------------------------------------------------------------------
_CORRECTORS = {}
_PREDICTORS = {}


def _registrar(table):
  """``@register(name=...)`` / ``@register`` decorator filling `table` (sampling.py:34-66)."""
  def register(cls=None, *, name=None):
    def add(c):
      key = c.__name__ if name is None else name
      if key in table:
        raise ValueError(f'Already registered model with name: {key}')
      table[key] = c
      return c
    return add if cls is None else add(cls)
  return register


register_predictor = _registrar(_PREDICTORS)
register_corrector = _registrar(_CORRECTORS)


def get_predictor(name):
  return _PREDICTORS[name]


def get_corrector(name):
  return _CORRECTORS[name]


class Predictor(abc.ABC):
  """Abstract predictor; builds the reverse SDE/ODE once (sampling.py:128-157)."""

  def __init__(self, sde, score_fn, probability_flow=False, logsnr_model=None):
    super().__init__()
    self.sde, self.score_fn = sde, score_fn
    if logsnr_model is not None:
      self.rsde = sde.reverse(score_fn, logsnr_model, probability_flow)
    else:
      self.rsde = sde.reverse(score_fn, probability_flow, lambda_=0. if probability_flow else 1.)

  @abc.abstractmethod
  def update_fn(self, x, t, next_t=None):
    """One predictor update -> (x, x_mean)."""


class Corrector(abc.ABC):
  """Abstract corrector (sampling.py:160-182)."""

  def __init__(self, sde, score_fn, snr, n_steps):
    super().__init__()
    self.sde, self.score_fn, self.snr, self.n_steps = sde, score_fn, snr, n_steps

  @abc.abstractmethod
  def update_fn(self, x, t):
    """One corrector update -> (x, x_mean)."""


@register_predictor(name='euler_maruyama')
class EulerMaruyamaPredictor(Predictor):
  def __init__(self, config, sde, score_fn, probability_flow=False):
    super().__init__(sde, score_fn, probability_flow)

  def update_fn(self, x, t):
    return _euler_maruyama(self.rsde, x, t)


@register_predictor(name='reverse_diffusion')
class ReverseDiffusionPredictor(Predictor):
  def __init__(self, config, sde, score_fn, probability_flow=False, logsnr_model=None):
    super().__init__(sde, score_fn, probability_flow, logsnr_model)
    self.config = config

  def update_fn(self, x, t, next_t=None):
    return _reverse_diffusion(self.rsde, x, t, next_t)


@register_predictor(name='ancestral_sampling')
class AncestralSamplingPredictor(Predictor):
  """Ancestral sampling for VE / VP SDEs (sampling.py:213-249)."""

  def __init__(self, config, sde, score_fn, probability_flow=False):
    super().__init__(sde, score_fn, probability_flow)
    if not isinstance(sde, (sde_lib.VPSDE, sde_lib.VESDE)):
      raise NotImplementedError(f"SDE class {sde.__class__.__name__} not yet supported.")
    assert not probability_flow, "Probability flow not supported by ancestral sampling"

  def vesde_update_fn(self, x, t):
    return _ancestral_ve(self.sde, self.score_fn, x, t)

  def vpsde_update_fn(self, x, t):
    return _ancestral_vp(self.sde, self.score_fn, x, t)

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


class _CheckedCorrector(Corrector):
  def __init__(self, sde, score_fn, snr, n_steps):
    super().__init__(sde, score_fn, snr, n_steps)
    if not isinstance(sde, (sde_lib.VPSDE, sde_lib.VESDE, sde_lib.subVPSDE)):
      raise NotImplementedError(f"SDE class {sde.__class__.__name__} not yet supported.")


@register_corrector(name='langevin')
class LangevinCorrector(_CheckedCorrector):
  """Langevin corrector with batch-mean norms (sampling.py:263-292)."""

  def update_fn(self, x, t):
    return _langevin(self.sde, self.score_fn, self.snr, self.n_steps, x, t)


@register_corrector(name='ald')
class AnnealedLangevinDynamics(_CheckedCorrector):
  """Annealed Langevin dynamics of NCSN/NCSNv2 (sampling.py:295-329)."""

  def update_fn(self, x, t):
    return _annealed_langevin(self.sde, self.score_fn, self.snr, self.n_steps, x, t)


@register_corrector(name='none')
class NoneCorrector(Corrector):
  """Identity corrector."""

  def __init__(self, sde, score_fn, snr, n_steps):
    pass

  def update_fn(self, x, t):
    return x, x


# ---- samplers -------------------------------------------------------------------------------------------------------
def shared_predictor_update_fn(x, t, sde, model, predictor, probability_flow, continuous, config):
  """Build the predictor for ``model`` and apply one update (sampling.py:343-351)."""
  score_fn = mutils.get_score_fn(config, sde, model, train=False, continuous=continuous)
  obj = NonePredictor(sde, score_fn, probability_flow) if predictor is None \
      else predictor(config, sde, score_fn, probability_flow)
  return obj.update_fn(x, t)


def shared_corrector_update_fn(x, t, sde, model, corrector, continuous, snr, n_steps, config):
  """Build the corrector for ``model`` and apply one update (sampling.py:354-362)."""
  score_fn = mutils.get_score_fn(config, sde, model, train=False, continuous=continuous)
  obj = NoneCorrector(sde, score_fn, snr, n_steps) if corrector is None else corrector(sde, score_fn, snr, n_steps)
  return obj.update_fn(x, t)


def _denoiser(config, sde, probability_flow):
  """Final noise-free reverse-diffusion step from ``sde.eps`` to 0, returning the mean (sampling.py:402-408,
  :457-463; the PC sampler uses the probability-flow form, the ODE sampler the SDE form)."""
  def denoise(model, x):
    score_fn = get_score_fn(config, sde, model, train=False, continuous=True)
    step = ReverseDiffusionPredictor(config, sde, score_fn, probability_flow=probability_flow)
    vec_eps = torch.ones(x.shape[0], device=x.device) * sde.eps
    return step.update_fn(x, vec_eps, torch.zeros_like(vec_eps))[1]
  return denoise


def get_sampling_fn(config, sde, shape, inverse_scaler, eps):
  """``sampling_fn(model) -> (samples, nfe)`` for ``config.sampling.method`` (sampling.py:80-125)."""
  s = config.sampling
  kind = s.method.lower()
  if kind == 'ode':
    return get_ode_sampler(config=config, sde=sde, shape=shape, inverse_scaler=inverse_scaler,
                           denoise=s.noise_removal, eps=eps, device=config.device)
  if kind == 'pc':
    return get_pc_sampler(config=config, sde=sde, shape=shape, predictor=get_predictor(s.predictor.lower()),
                          corrector=get_corrector(s.corrector.lower()), inverse_scaler=inverse_scaler, snr=s.snr,
                          n_steps=s.n_steps_each, probability_flow=s.probability_flow,
                          continuous=config.training.continuous, denoise=s.noise_removal, eps=eps,
                          device=config.device)
  raise ValueError(f"Sampler name {s.method} unknown.")


def get_pc_sampler(config, sde, shape, predictor, corrector, inverse_scaler, snr, n_steps=1,
                   probability_flow=False, continuous=False, denoise=True, eps=1e-3, device='cuda'):
  """Predictor-corrector sampler (sampling.py:365-433): at each of the N times from T down to eps the corrector runs
  first, then the predictor; ``nfe = N (n_steps + 1)``."""
  predict = functools.partial(shared_predictor_update_fn, sde=sde, predictor=predictor,
                              probability_flow=probability_flow, continuous=continuous, config=config)
  correct = functools.partial(shared_corrector_update_fn, sde=sde, corrector=corrector, continuous=continuous, snr=snr,
                              n_steps=n_steps, config=config)
  denoise_update_fn = _denoiser(config, sde, probability_flow=True)

  def pc_sampler(model):
    # the parameters are fixed for the whole loop: convolution weights are prepared once (models.utils.frozen_weights)
    with torch.no_grad(), mutils.frozen_weights(model):
      x = sde.prior_sampling(shape).to(device)
      grid = torch.linspace(sde.T, eps, sde.N, device=device)
      for i in tqdm(range(sde.N)):
        vec_t = torch.ones(shape[0], device=grid.device) * grid[i]
        x, x_mean = correct(x, vec_t, model=model)
        x, x_mean = predict(x, vec_t, model=model)
      x_mean = x = denoise_update_fn(model, x_mean if denoise else x)
      return inverse_scaler(x_mean if denoise else x), sde.N * (n_steps + 1)

  return pc_sampler


def get_ode_sampler(config, sde, shape, inverse_scaler, denoise=False, rtol=1e-5, atol=1e-5,
                    method='RK45', eps=1e-3, device='cuda'):
  """Probability-flow ODE sampler (sampling.py:436-504): SciPy's RK45 algorithm, float64 solver state.

  `method='RK45'` (every config) runs the solver on device tensors (engine/rk45.py: same tableau, error norm and step
  controller, checked against SciPy itself), so the state stays on the device instead of making a host round trip per
  network evaluation; the network sees the same float32 cast of it as in the reference.  Other methods go through
  scipy.integrate.solve_ivp on the host exactly as the reference does."""
  denoise_update_fn = _denoiser(config, sde, probability_flow=False)

  def drift_fn(model, x, t):
    score_fn = get_score_fn(config, sde, model, train=False, continuous=True)
    return sde.reverse(score_fn, probability_flow=True, lambda_=0.).sde(x, t)[0]

  def drift_at(model, t, x):
    return drift_fn(model, x, torch.ones(shape[0], device=x.device) * t)

  def integrate_on_device(model, x):
    def ode_func(t, y):
      return drift_at(model, t, y.reshape(shape).to(torch.float32)).reshape(-1).to(torch.float64)
    y, nfe = rk45.solve_ivp_rk45(ode_func, (sde.T, eps), x.reshape(-1).to(torch.float64), rtol=rtol, atol=atol)
    return y.reshape(shape), nfe

  def integrate_on_host(model, x):
    def ode_func(t, flat):
      xt = from_flattened_numpy(flat, shape).to(device).type(torch.float32)
      return to_flattened_numpy(drift_at(model, t, xt))
    solution = integrate.solve_ivp(ode_func, (sde.T, eps), to_flattened_numpy(x), rtol=rtol, atol=atol, method=method)
    return torch.tensor(solution.y[:, -1]).reshape(shape), solution.nfev

  def ode_sampler(model):
    with torch.no_grad(), mutils.frozen_weights(model):
      x = sde.prior_sampling(shape).to(device)
      x, nfe = (integrate_on_device if method == 'RK45' else integrate_on_host)(model, x)
      x = x.to(device).type(torch.float32)
      if denoise:
        x = denoise_update_fn(model, x)
      return inverse_scaler(x), nfe

  return ode_sampler
