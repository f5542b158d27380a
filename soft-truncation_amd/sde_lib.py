"""SDE algebra of the hot path: forward SDEs, their time reversal, and the Soft-Truncation time sampling.

Interface parity with the reference ``sde_lib.py`` (file:line = /root/reference/sde_lib.py): abstract ``SDE``
(:8-119) with ``T, sde, marginal_prob, prior_sampling, prior_logp, discretize, reverse``; ``VPSDE`` (:121-207),
``subVPSDE`` (:209-246), ``VESDE`` (:248-332), ``reciprocal_VESDE`` (:334-430); factory ``get_sde`` (:433-445); the
Soft-Truncation additions ``integral_beta / antiderivative / normalizing_constant / get_diffusion_time / get_t_min``.

How this file is organised (it is not laid out like the reference): the pieces several SDEs share live once, in
small mixins -- unit time horizon, isotropic Gaussian prior of a given scale, the geometric sigma ladder, the
``A(T) - A(t_min)`` normaliser of the importance-sampled time -- and the noise schedules are module-level functions of
``t``.  What must NOT change is the *order of floating-point operations* inside each formula: everything here is
per-sample scalar math on ``t[B]`` plus one broadcast multiply on the image tensor, and tests/test_oracle_golden.py pins
the CPU results bit-for-bit to fixtures generated from the reference.  The heavy tensor work of the path (the score
network) lives in ``models/`` and the HIP engine.

Reference quirks that are deliberately reproduced: ``VESDE.get_t_min`` /
``reciprocal_VESDE.get_t_min`` default ``st=False`` so the training step always gets ``eps``; RVE ``marginal_prob``
computes in float64 on the host then casts; ``reciprocal_VESDE.discretize`` needs ``next_t``.
"""
import abc

import numpy as np
import torch


def _bcast(v):
  """[B] -> [B,1,1,1] view for image-shaped broadcasting."""
  return v[:, None, None, None]


def _as_time_tensor(t):
  return torch.tensor(t).float() if isinstance(t, (float, int)) else t


# ---- noise schedules --------------------------------------------------------------------------------------------
def _linear_beta(b0, b1, t):
  """beta(t) of the VP family."""
  return b0 + t * (b1 - b0)


def _vp_log_mean_coeff(b0, b1, t):
  """log of the mean coefficient of p_t(x_t | x_0) for linear beta (sde_lib.py:152, :232)."""
  return -0.25 * t ** 2 * (b1 - b0) - 0.5 * t * b0


def _geometric_sigma(lo, hi, t):
  """sigma(t) = lo (hi / lo)^t of the VE family (sde_lib.py:276)."""
  return lo * (hi / lo) ** t


def _sigma_ladder(lo, hi, n):
  """The n discrete noise levels of SMLD, geometric between lo and hi (sde_lib.py:262, :372)."""
  return torch.exp(torch.linspace(np.log(lo), np.log(hi), n))


def _log_uniform_t_min(eps, k):
  """Soft-Truncation bound drawn from numpy's global stream (sde_lib.py:200-207): log-uniform on [eps, 1] for k == 1,
  otherwise eps / (1 - u (1 - eps^(k-1)))^(1/(k-1))."""
  u = np.random.rand()
  if k == 1.0:
    return eps ** (1. - u)
  return eps / (1. - u * (1 - eps ** (k - 1))) ** (1. / (k - 1))


def _uniform_time(n, device, lo, hi):
  return torch.rand(n, device=device) * (hi - lo) + lo


# ---- the abstract interface and the time reversal ---------------------------------------------------------------
class SDE(abc.ABC):
  """Abstract forward SDE  dx = f(x,t) dt + g(t) dw  on mini-batches (sde_lib.py:8)."""

  def __init__(self, N):
    super().__init__()
    self.N = N

  @property
  @abc.abstractmethod
  def T(self):
    """End time of the SDE."""

  @abc.abstractmethod
  def sde(self, x, t):
    """Drift and diffusion (f(x,t), g(t))."""

  @abc.abstractmethod
  def marginal_prob(self, x, t):
    """Mean and std of p_t(x_t | x_0 = x)."""

  @abc.abstractmethod
  def prior_sampling(self, shape):
    """One sample of the prior p_T."""

  @abc.abstractmethod
  def prior_logp(self, z):
    """log p_T(z)."""

  def get_diffusion_time(self, config):
    pass

  def discretize(self, x, t, next_t=None):
    """Euler-Maruyama default: x_{i+1} = x_i + f_i + G_i z  (sde_lib.py:56-73)."""
    step = 1 / self.N
    f, g = self.sde(x, t)
    return f * step, g * torch.sqrt(torch.tensor(step, device=t.device))

  def reverse(self, score_fn, probability_flow=False, lambda_=1.):
    """Reverse-time SDE / probability-flow ODE (sde_lib.py:75-119): an instance of a subclass of ``type(self)``
    exposing ``sde`` and ``discretize`` of the reverse process.  The score enters with weight 1/2 for the ODE and
    (1 + lambda^2)/2 otherwise; the diffusion is scaled by lambda."""
    assert probability_flow == (lambda_ == 0.)
    return _reverse_of(self, score_fn, probability_flow, lambda_)


def _reverse_of(forward, score_fn, probability_flow, lambda_):
  score_weight = 0.5 if probability_flow else 0.5 * (1. + lambda_ ** 2)

  def against_score(drift, g, x, t):
    return drift - _bcast(g) ** 2 * score_fn(x, t) * score_weight, lambda_ * g

  class RSDE(type(forward)):
    def __init__(self):             # deliberately not the forward class' constructor: state is borrowed from `forward`
      self.N = forward.N
      self.probability_flow = probability_flow
      self.lambda_ = lambda_
      self.weight = score_weight

    T = property(lambda self: forward.T)

    def sde(self, x, t):
      drift, diffusion = forward.sde(x, t)
      return against_score(drift, diffusion, x, t)

    def discretize(self, x, t, next_t=None):
      f, G = forward.discretize(x, t, next_t)
      return against_score(f, G, x, t)

  return RSDE()


# ---- shared traits --------------------------------------------------------------------------------------------------
class _UnitHorizon:
  @property
  def T(self):
    return 1


class _GaussianPrior:
  """p_T = N(0, s^2 I) with s = ``self._prior_scale`` (1 for the VP family, sigma_max for the VE family)."""
  _prior_scale = 1

  def prior_sampling(self, shape, data_mean=None):
    z = torch.randn(*shape)
    return z if self._prior_scale == 1 else z * self._prior_scale

  def prior_logp(self, z):
    n, s = np.prod(z.shape[1:]), self._prior_scale
    sq = torch.sum(z ** 2, dim=(1, 2, 3))
    if s == 1:
      return -n / 2. * np.log(2 * np.pi) - sq / 2.
    return -n / 2. * np.log(2 * np.pi * s ** 2) - sq / (2 * s ** 2)


class _ImportanceTime:
  """Z(t_min) = A(T) - A(t_min): the mass of the importance-sampled diffusion time (sde_lib.py:188, :311)."""

  def normalizing_constant(self, t_min):
    return self.antiderivative(self.T) - self.antiderivative(t_min)


# ---- VP family ------------------------------------------------------------------------------------------------------
class VPSDE(_UnitHorizon, _GaussianPrior, _ImportanceTime, SDE):
  """Variance-preserving SDE (sde_lib.py:121-207)."""

  def __init__(self, truncation_time=1e-5, beta_min=0.1, beta_max=20, N=1000):
    super().__init__(N)
    self.beta_0, self.beta_1 = beta_min, beta_max
    self.eps = truncation_time
    # DDPM ladder (buffers the discrete samplers index)
    self.discrete_betas = torch.linspace(beta_min / N, beta_max / N, N)
    self.alphas = 1. - self.discrete_betas
    self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
    self.sqrt_alphas_cumprod = torch.sqrt(self.alphas_cumprod)
    self.sqrt_1m_alphas_cumprod = torch.sqrt(1. - self.alphas_cumprod)

  def sde(self, x, t):
    beta_t = _linear_beta(self.beta_0, self.beta_1, t)
    return -0.5 * _bcast(beta_t) * x, torch.sqrt(beta_t)

  def marginal_prob(self, x, t):
    lmc = _vp_log_mean_coeff(self.beta_0, self.beta_1, t)
    return torch.exp(_bcast(lmc)) * x, torch.sqrt(1. - torch.exp(2. * lmc))

  def discretize(self, x, t, next_t=None):
    """DDPM ladder, or the continuous (t -> next_t) form (sde_lib.py:166-178)."""
    if next_t is not None:
      G = torch.sqrt((t - next_t) * (self.beta_0 + (self.beta_1 - self.beta_0) * t))
      return _bcast(torch.sqrt(1. - G ** 2)) * x - x, G
    idx = (t * (self.N - 1) / self.T).long()
    beta, alpha = self.discrete_betas.to(x.device)[idx], self.alphas.to(x.device)[idx]
    return _bcast(torch.sqrt(alpha)) * x - x, torch.sqrt(beta)

  # Soft Truncation
  def integral_beta(self, t):
    return 0.5 * t ** 2 * (self.beta_1 - self.beta_0) + t * self.beta_0

  def antiderivative(self, t, stabilizing_constant=0.):
    """A(t) = log(1 - exp(-B(t)) + c) + B(t), the antiderivative of g^2/sigma^2 (:183-186)."""
    t = _as_time_tensor(t)
    return torch.log(1. - torch.exp(- self.integral_beta(t)) + stabilizing_constant) + self.integral_beta(t)

  def get_diffusion_time(self, config, batch_size, batch_device, t_min, importance_sampling=True):
    """Inverse-CDF sample of t proportional to g^2/sigma^2 on [t_min, 1] (:191-198); else uniform with Z = 1."""
    if not importance_sampling:
      return _uniform_time(batch_size, batch_device, t_min, self.T), 1
    Z = self.normalizing_constant(t_min)
    u = torch.rand(batch_size, device=batch_device)
    slope = self.beta_1 - self.beta_0
    t = (-self.beta_0 + torch.sqrt(self.beta_0 ** 2 + 2 * slope *
                                   torch.log(1. + torch.exp(Z * u + self.antiderivative(t_min))))) / slope
    return t, Z.detach()

  def get_t_min(self, config):
    return _log_uniform_t_min(self.eps, config.training.k) if config.training.st else self.eps


class subVPSDE(_UnitHorizon, _GaussianPrior, SDE):
  """sub-VP SDE (sde_lib.py:209-246).  No Soft-Truncation time sampler in the reference, so it cannot drive
  ``losses.get_step_fn``; kept for interface completeness."""

  def __init__(self, truncation_time=1e-5, beta_min=0.1, beta_max=20, N=1000):
    super().__init__(N)
    self.beta_0, self.beta_1 = beta_min, beta_max

  def sde(self, x, t):
    beta_t = _linear_beta(self.beta_0, self.beta_1, t)
    discount = 1. - torch.exp(-2 * self.beta_0 * t - (self.beta_1 - self.beta_0) * t ** 2)
    return -0.5 * _bcast(beta_t) * x, torch.sqrt(beta_t * discount)

  def marginal_prob(self, x, t):
    lmc = _vp_log_mean_coeff(self.beta_0, self.beta_1, t)
    return _bcast(torch.exp(lmc)) * x, 1 - torch.exp(2. * lmc)


# ---- VE family ------------------------------------------------------------------------------------------------------
class VESDE(_UnitHorizon, _GaussianPrior, _ImportanceTime, SDE):
  """Variance-exploding SDE (sde_lib.py:248-332)."""

  def __init__(self, sigma_min=0.01, sigma_max=50, N=1000, truncation_time=1e-5):
    super().__init__(N)
    self.sigma_min, self.sigma_max = sigma_min, sigma_max
    self._prior_scale = sigma_max
    self.eps = truncation_time
    self.discrete_sigmas = _sigma_ladder(sigma_min, sigma_max, N)

  def _sigma(self, t):
    return _geometric_sigma(self.sigma_min, self.sigma_max, t)

  def _log_ratio(self):
    return np.log(self.sigma_max) - np.log(self.sigma_min)

  def sde(self, x, t):
    g = self._sigma(t) * torch.sqrt(torch.tensor(2 * self._log_ratio(), device=t.device))
    return torch.zeros_like(x), g

  def marginal_prob(self, x, t):
    return x, self._sigma(t)

  def discretize(self, x, t, next_t=None):
    """SMLD ladder; ``next_t == 0`` is the denoising special case (sde_lib.py:288-304)."""
    if next_t is None:
      idx = (t * (self.N - 1) / self.T).long()
      ladder = self.discrete_sigmas.to(t.device)
      sigma, below = ladder[idx], torch.where(idx == 0, torch.zeros_like(t), ladder[idx - 1])
    elif next_t[0].item() == 0.:
      sigma, below = self._sigma(t), self._sigma(next_t)
    else:
      raise NotImplementedError
    return torch.zeros_like(x), torch.sqrt(sigma ** 2 - below ** 2)

  def antiderivative(self, t):
    return 2. * torch.log(self._sigma(_as_time_tensor(t)))

  def get_diffusion_time(self, config, batch_size, batch_device, t_min, importance_sampling=None):
    if importance_sampling is None:
      importance_sampling = config.training.importance_sampling
    if not importance_sampling:
      return _uniform_time(batch_size, batch_device, t_min, self.T), 1
    Z = self.normalizing_constant(t_min)
    u = torch.rand(batch_size, device=batch_device)
    return t_min + ((Z * u) / (2. * self._log_ratio())), Z.detach()

  def get_t_min(self, config, st=False):
    # NB: callers pass no `st`, so this is `eps` in every live training step (sde_lib.py:324).
    return _log_uniform_t_min(self.eps, config.training.k) if st else self.eps


class reciprocal_VESDE(_UnitHorizon, _GaussianPrior, SDE):
  """Reciprocal VE SDE, sigma(t)^2 = c b^(2/t) + c2 b2^(2/t)  (sde_lib.py:334-430)."""

  def __init__(self, eta=1e-5, sigma_min=0.01, sigma_max=50, N=1000):
    super().__init__(N)
    self.sigma_min, self.sigma_max, self.eta = sigma_min, sigma_max, eta
    self._prior_scale = sigma_max
    self.eps = 1e-5
    horizon = 1. / self.eps
    # the two exponential components of sigma(t)^2
    self.base_sigma = pow(self.eta / self.sigma_max, 1. / ((horizon - 1.)))
    self.const = self.sigma_max ** 2 / self.base_sigma ** 2
    self.base_sigma_2 = pow(1.01, - 1. / (2. * (horizon - 1.)))
    self.const_2 = - pow(1.01, (horizon) / (horizon - 1.)) * (self.eta ** 2 - self.sigma_min ** 2)
    # constants of `transform` (a log-sigma conditioning the reference defines but never reaches)
    self.t_0 = torch.tensor(self.get_time())
    self.sigma_0 = torch.sqrt(self._variance(self.t_0, exponent=lambda t: 2. * t))
    log_b = np.log(self.base_sigma)
    self.k_1 = - self.t_0 * self.sigma_0 / log_b
    self.k_2 = - self.k_1 / self.sigma_0
    self.constant_ = 1. / torch.log(self.sigma_0 / self.sigma_max)
    self.c_1_ = self.sigma_0 / log_b * (np.log(self.sigma_0) - np.log(self.sigma_max)) / (self.t_0 - 1. / self.T)
    self.c_2_ = self.sigma_0 - (self.c_1_ / self.sigma_0)
    self.c_2__ = np.log(self.sigma_0) + self.c_1_ / self.sigma_0
    self.discrete_sigmas = _sigma_ladder(sigma_min, sigma_max, N)

  def _variance(self, t, exponent=lambda t: 2. / t):
    e = exponent(t)
    return self.const * torch.pow(self.base_sigma, e) + self.const_2 * torch.pow(self.base_sigma_2, e)

  def sde(self, x, t):
    first = -(2. * self.const * np.log(self.base_sigma)) * torch.pow(self.base_sigma, 2. / t) / (t ** 2)
    second = (2. * self.const_2 * np.log(self.base_sigma_2) * torch.pow(self.base_sigma_2, 2. / t) / (t ** 2))
    return torch.zeros_like(x), torch.sqrt(first + second)

  def marginal_prob(self, x, t):
    # float64 on the host, cast back to f32 on x's device (sde_lib.py:381-385)
    std = torch.sqrt(self._variance(t.type(torch.DoubleTensor)))
    return x, std.type(torch.float32).to(x.device)

  def prior_sampling(self, shape):
    return torch.randn(*shape) * self.sigma_max

  def discretize(self, x, t, next_t=None):
    sigma = self.marginal_prob(x, t)[1]
    below = next_t if next_t.type == 'torch.IntTensor' else self.marginal_prob(x, next_t)[1]
    return torch.zeros_like(x), torch.sqrt(sigma ** 2 - below ** 2)

  def get_time(self, sigma_level=0.01):
    return np.log((-self.sigma_min ** 2 + self.eta ** 2 + sigma_level ** 2) / self.const) \
           / (2. * np.log(self.base_sigma))

  def transform(self, sigmas):
    return (sigmas > 0.01) * torch.log(sigmas) + (sigmas < 0.01) * (-self.c_1_ / (sigmas + 1e-4) + self.c_2__)

  def get_diffusion_time(self, config, batch_size, batch_device, t_min, importance_sampling=False):
    """t = 1 / U[1/T, 1/t_min], Z = 1 (sde_lib.py:421-423)."""
    inverse = torch.rand(batch_size, device=batch_device) * (1. / t_min - 1. / self.T) + 1. / self.T
    return 1. / inverse, 1

  def get_t_min(self, config, st=False):
    if not st:
      return self.eps
    return 1. / (np.random.rand() * (1. / self.eps - 1. / self.T) + 1. / self.T)


# ---- factory --------------------------------------------------------------------------------------------------------
def get_sde(config, state):
  """Factory keyed on ``config.training.sde`` (sde_lib.py:433-445)."""
  kind, m, tr = config.training.sde.lower(), config.model, config.training
  if kind in ('vpsde', 'subvpsde'):
    cls = VPSDE if kind == 'vpsde' else subVPSDE
    return cls(truncation_time=tr.truncation_time, beta_min=m.beta_min, beta_max=m.beta_max, N=m.num_scales)
  if kind == 'vesde':
    return VESDE(sigma_min=m.sigma_min, sigma_max=m.sigma_max, N=m.num_scales)
  if kind == 'reciprocal_vesde':
    return reciprocal_VESDE(sigma_min=m.sigma_min, sigma_max=m.sigma_max, N=m.num_scales, eta=tr.eta)
  raise NotImplementedError(f"SDE {config.training.sde} unknown.")
