"""SDE algebra of the hot path: forward SDEs, their reverse, and the Soft-Truncation
time sampling.

Interface parity with the reference ``sde_lib.py`` (file:line refer to
/root/reference/sde_lib.py):

* abstract ``SDE``  (:8-119): ``T, sde, marginal_prob, prior_sampling, prior_logp,
  discretize, reverse``;
* ``VPSDE`` (:121-207), ``subVPSDE`` (:209-246), ``VESDE`` (:248-332),
  ``reciprocal_VESDE`` (:334-430), factory ``get_sde`` (:433-445);
* Soft-Truncation additions: ``integral_beta``/``antiderivative``/
  ``normalizing_constant``/``get_diffusion_time``/``get_t_min``.

Everything here is per-sample scalar math on ``t[B]`` plus one broadcast multiply
on the image tensor; the arithmetic *order* of every expression follows the
reference so the CPU results are bit-identical (tests/test_sde_parity.py pins that
against fixtures generated from the reference).  The heavy tensor work of the path
(the score network) lives in ``models/`` and the HIP engine.

Reference quirks that are deliberately reproduced:
``VESDE.get_t_min`` / ``reciprocal_VESDE.get_t_min`` default ``st=False`` so the
training step always gets ``eps``; RVE ``marginal_prob`` computes in float64 on the
host then casts; ``reciprocal_VESDE.discretize`` needs ``next_t``.
"""
import abc

import numpy as np
import torch


def _bcast(v):
  """[B] -> [B,1,1,1] view for image-shaped broadcasting."""
  return v[:, None, None, None]


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
    dt = 1 / self.N
    drift, diffusion = self.sde(x, t)
    f = drift * dt
    G = diffusion * torch.sqrt(torch.tensor(dt, device=t.device))
    return f, G

  def reverse(self, score_fn, probability_flow=False, lambda_=1.):
    """Reverse-time SDE / probability-flow ODE (sde_lib.py:75-119).

    The returned object is an instance of a subclass of ``type(self)`` exposing
    ``sde`` and ``discretize`` of the reverse process; ``weight`` is 1/2 for the ODE and
    (1+lambda^2)/2 otherwise.
    """
    assert probability_flow == (lambda_ == 0.)
    fwd = self
    weight = 0.5 if probability_flow else 0.5 * (1. + lambda_ ** 2)

    class RSDE(fwd.__class__):
      def __init__(self):
        self.N = fwd.N
        self.probability_flow = probability_flow
        self.lambda_ = lambda_
        self.weight = weight

      @property
      def T(self):
        return fwd.T

      def sde(self, x, t):
        drift, diffusion = fwd.sde(x, t)
        score = score_fn(x, t)
        drift = drift - _bcast(diffusion) ** 2 * score * self.weight
        diffusion = self.lambda_ * diffusion
        return drift, diffusion

      def discretize(self, x, t, next_t=None):
        f, G = fwd.discretize(x, t, next_t)
        rev_f = f - _bcast(G) ** 2 * score_fn(x, t) * self.weight
        rev_G = self.lambda_ * G
        return rev_f, rev_G

    return RSDE()


def _soft_truncation_t_min(eps, k):
  """Draw the per-step truncation bound from the numpy global stream (sde_lib.py:200-207).

  k == 1: log-uniform on [eps, 1];  otherwise eps / (1 - u (1 - eps^(k-1)))^(1/(k-1)).
  """
  if k == 1.0:
    return eps ** (1. - np.random.rand())
  return eps / (1. - np.random.rand() * (1 - eps ** (k - 1))) ** (1. / (k - 1))


class VPSDE(SDE):
  """Variance-preserving SDE (sde_lib.py:121-207)."""

  def __init__(self, truncation_time=1e-5, beta_min=0.1, beta_max=20, N=1000):
    super().__init__(N)
    self.beta_0 = beta_min
    self.beta_1 = beta_max
    self.eps = truncation_time
    self.N = N
    self.discrete_betas = torch.linspace(beta_min / N, beta_max / N, N)
    self.alphas = 1. - self.discrete_betas
    self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
    self.sqrt_alphas_cumprod = torch.sqrt(self.alphas_cumprod)
    self.sqrt_1m_alphas_cumprod = torch.sqrt(1. - self.alphas_cumprod)

  @property
  def T(self):
    return 1

  def sde(self, x, t):
    beta_t = self.beta_0 + t * (self.beta_1 - self.beta_0)
    drift = -0.5 * _bcast(beta_t) * x
    diffusion = torch.sqrt(beta_t)
    return drift, diffusion

  def marginal_prob(self, x, t):
    log_mean_coeff = -0.25 * t ** 2 * (self.beta_1 - self.beta_0) - 0.5 * t * self.beta_0
    mean = torch.exp(_bcast(log_mean_coeff)) * x
    std = torch.sqrt(1. - torch.exp(2. * log_mean_coeff))
    return mean, std

  def prior_sampling(self, shape):
    return torch.randn(*shape)

  def prior_logp(self, z):
    n = np.prod(z.shape[1:])
    return -n / 2. * np.log(2 * np.pi) - torch.sum(z ** 2, dim=(1, 2, 3)) / 2.

  def discretize(self, x, t, next_t=None):
    """DDPM ladder, or the continuous (t -> next_t) form (sde_lib.py:166-178)."""
    if next_t is None:
      timestep = (t * (self.N - 1) / self.T).long()
      beta = self.discrete_betas.to(x.device)[timestep]
      alpha = self.alphas.to(x.device)[timestep]
      sqrt_beta = torch.sqrt(beta)
      f = _bcast(torch.sqrt(alpha)) * x - x
      G = sqrt_beta
    else:
      G = torch.sqrt((t - next_t) * (self.beta_0 + (self.beta_1 - self.beta_0) * t))
      f = _bcast(torch.sqrt(1. - G ** 2)) * x - x
    return f, G

  # ---- Soft Truncation -------------------------------------------------------------
  def integral_beta(self, t):
    return 0.5 * t ** 2 * (self.beta_1 - self.beta_0) + t * self.beta_0

  def antiderivative(self, t, stabilizing_constant=0.):
    """A(t) = log(1 - exp(-B(t)) + c) + B(t), the antiderivative of g^2/sigma^2 (:183-186)."""
    if isinstance(t, (float, int)):
      t = torch.tensor(t).float()
    return torch.log(1. - torch.exp(- self.integral_beta(t)) + stabilizing_constant) + self.integral_beta(t)

  def normalizing_constant(self, t_min):
    return self.antiderivative(self.T) - self.antiderivative(t_min)

  def get_diffusion_time(self, config, batch_size, batch_device, t_min, importance_sampling=True):
    """Inverse-CDF sample of t proportional to g^2/sigma^2 on [t_min, 1] (:191-198)."""
    if importance_sampling:
      Z = self.normalizing_constant(t_min)
      u = torch.rand(batch_size, device=batch_device)
      t = (-self.beta_0 + torch.sqrt(self.beta_0 ** 2 + 2 * (self.beta_1 - self.beta_0) *
                                     torch.log(1. + torch.exp(Z * u + self.antiderivative(t_min))))) \
          / (self.beta_1 - self.beta_0)
      return t, Z.detach()
    return torch.rand(batch_size, device=batch_device) * (self.T - t_min) + t_min, 1

  def get_t_min(self, config):
    if config.training.st:
      return _soft_truncation_t_min(self.eps, config.training.k)
    return self.eps


class subVPSDE(SDE):
  """sub-VP SDE (sde_lib.py:209-246).  No Soft-Truncation time sampler in the reference,
  so it cannot drive ``losses.get_step_fn``; kept for interface completeness."""

  def __init__(self, truncation_time=1e-5, beta_min=0.1, beta_max=20, N=1000):
    super().__init__(N)
    self.beta_0 = beta_min
    self.beta_1 = beta_max
    self.N = N

  @property
  def T(self):
    return 1

  def sde(self, x, t):
    beta_t = self.beta_0 + t * (self.beta_1 - self.beta_0)
    drift = -0.5 * _bcast(beta_t) * x
    discount = 1. - torch.exp(-2 * self.beta_0 * t - (self.beta_1 - self.beta_0) * t ** 2)
    diffusion = torch.sqrt(beta_t * discount)
    return drift, diffusion

  def marginal_prob(self, x, t):
    log_mean_coeff = -0.25 * t ** 2 * (self.beta_1 - self.beta_0) - 0.5 * t * self.beta_0
    mean = _bcast(torch.exp(log_mean_coeff)) * x
    std = 1 - torch.exp(2. * log_mean_coeff)
    return mean, std

  def prior_sampling(self, shape, data_mean=None):
    return torch.randn(*shape)

  def prior_logp(self, z):
    n = np.prod(z.shape[1:])
    return -n / 2. * np.log(2 * np.pi) - torch.sum(z ** 2, dim=(1, 2, 3)) / 2.


class VESDE(SDE):
  """Variance-exploding SDE (sde_lib.py:248-332)."""

  def __init__(self, sigma_min=0.01, sigma_max=50, N=1000, truncation_time=1e-5):
    super().__init__(N)
    self.sigma_min = sigma_min
    self.sigma_max = sigma_max
    self.eps = truncation_time
    self.discrete_sigmas = torch.exp(torch.linspace(np.log(self.sigma_min), np.log(self.sigma_max), N))
    self.N = N

  @property
  def T(self):
    return 1

  def _sigma(self, t):
    return self.sigma_min * (self.sigma_max / self.sigma_min) ** t

  def sde(self, x, t):
    sigma = self._sigma(t)
    drift = torch.zeros_like(x)
    diffusion = sigma * torch.sqrt(torch.tensor(2 * (np.log(self.sigma_max) - np.log(self.sigma_min)),
                                                device=t.device))
    return drift, diffusion

  def marginal_prob(self, x, t):
    return x, self._sigma(t)

  def prior_sampling(self, shape):
    return torch.randn(*shape) * self.sigma_max

  def prior_logp(self, z):
    n = np.prod(z.shape[1:])
    return -n / 2. * np.log(2 * np.pi * self.sigma_max ** 2) \
           - torch.sum(z ** 2, dim=(1, 2, 3)) / (2 * self.sigma_max ** 2)

  def discretize(self, x, t, next_t=None):
    """SMLD ladder; ``next_t == 0`` is the denoising special case (sde_lib.py:288-304)."""
    if next_t is None:
      timestep = (t * (self.N - 1) / self.T).long()
      sigma = self.discrete_sigmas.to(t.device)[timestep]
      adjacent_sigma = torch.where(timestep == 0, torch.zeros_like(t),
                                   self.discrete_sigmas.to(t.device)[timestep - 1])
    else:
      if next_t[0].item() == 0.:
        sigma = self._sigma(t)
        adjacent_sigma = self._sigma(next_t)
      else:
        raise NotImplementedError
    f = torch.zeros_like(x)
    G = torch.sqrt(sigma ** 2 - adjacent_sigma ** 2)
    return f, G

  def antiderivative(self, t):
    if isinstance(t, (float, int)):
      t = torch.tensor(t).float()
    return 2. * torch.log(self._sigma(t))

  def normalizing_constant(self, t_min):
    return self.antiderivative(self.T) - self.antiderivative(t_min)

  def get_diffusion_time(self, config, batch_size, batch_device, t_min, importance_sampling=None):
    if importance_sampling is None:
      importance_sampling = config.training.importance_sampling
    if importance_sampling:
      Z = self.normalizing_constant(t_min)
      u = torch.rand(batch_size, device=batch_device)
      return t_min + ((Z * u) / (2. * (np.log(self.sigma_max) - np.log(self.sigma_min)))), Z.detach()
    return torch.rand(batch_size, device=batch_device) * (self.T - t_min) + t_min, 1

  def get_t_min(self, config, st=False):
    # NB: callers pass no `st`, so this is `eps` in every live training step (sde_lib.py:324).
    if st:
      return _soft_truncation_t_min(self.eps, config.training.k)
    return self.eps


class reciprocal_VESDE(SDE):
  """Reciprocal VE SDE, sigma(t)^2 = c b^(2/t) + c2 b2^(2/t)  (sde_lib.py:334-430)."""

  def __init__(self, eta=1e-5, sigma_min=0.01, sigma_max=50, N=1000):
    super().__init__(N)
    self.sigma_min = sigma_min
    self.sigma_max = sigma_max
    self.eta = eta
    self.eps = 1e-5
    inv = 1. / self.eps
    self.base_sigma = pow(self.eta / self.sigma_max, 1. / ((inv - 1.)))
    self.const = self.sigma_max ** 2 / self.base_sigma ** 2
    self.base_sigma_2 = pow(1.01, - 1. / (2. * (inv - 1.)))
    self.const_2 = - pow(1.01, (inv) / (inv - 1.)) * (self.eta ** 2 - self.sigma_min ** 2)

    self.t_0 = torch.tensor(self.get_time())
    self.sigma_0 = torch.sqrt(
      self.const * torch.pow(self.base_sigma, 2. * self.t_0)
      + self.const_2 * torch.pow(self.base_sigma_2, 2. * self.t_0))
    self.k_1 = - self.t_0 * self.sigma_0 / np.log(self.base_sigma)
    self.k_2 = - self.k_1 / self.sigma_0
    self.constant_ = 1. / torch.log(self.sigma_0 / self.sigma_max)
    self.c_1_ = self.sigma_0 / np.log(self.base_sigma) * (np.log(self.sigma_0) - np.log(self.sigma_max)) \
                / (self.t_0 - 1. / self.T)
    self.c_2_ = self.sigma_0 - (self.c_1_ / self.sigma_0)
    self.c_2__ = np.log(self.sigma_0) + self.c_1_ / self.sigma_0

    self.discrete_sigmas = torch.exp(torch.linspace(np.log(self.sigma_min), np.log(self.sigma_max), N))
    self.N = N

  @property
  def T(self):
    return 1

  def sde(self, x, t):
    drift = torch.zeros_like(x)
    diffusion = torch.sqrt(
      -(2. * self.const * np.log(self.base_sigma)) * torch.pow(self.base_sigma, 2. / t) / (t ** 2)
      + (2. * self.const_2 * np.log(self.base_sigma_2) * torch.pow(self.base_sigma_2, 2. / t) / (t ** 2)))
    return drift, diffusion

  def marginal_prob(self, x, t):
    # float64 on the host, cast back to f32 on x's device (sde_lib.py:381-385).
    t = t.type(torch.DoubleTensor)
    std = torch.sqrt(self.const * torch.pow(self.base_sigma, 2. / t)
                     + self.const_2 * torch.pow(self.base_sigma_2, 2. / t))
    return x, std.type(torch.float32).to(x.device)

  def prior_sampling(self, shape):
    return torch.randn(*shape) * self.sigma_max

  def prior_logp(self, z):
    n = np.prod(z.shape[1:])
    return -n / 2. * np.log(2 * np.pi * self.sigma_max ** 2) \
           - torch.sum(z ** 2, dim=(1, 2, 3)) / (2 * self.sigma_max ** 2)

  def discretize(self, x, t, next_t=None):
    sigma = self.marginal_prob(x, t)[1]
    if next_t.type == 'torch.IntTensor':
      next_sigma = next_t
    else:
      next_sigma = self.marginal_prob(x, next_t)[1]
    f = torch.zeros_like(x)
    G = torch.sqrt(sigma ** 2 - next_sigma ** 2)
    return f, G

  def get_time(self, sigma_level=0.01):
    return np.log((-self.sigma_min ** 2 + self.eta ** 2 + sigma_level ** 2) / self.const) \
           / (2. * np.log(self.base_sigma))

  def transform(self, sigmas):
    return (sigmas > 0.01) * torch.log(sigmas) + (sigmas < 0.01) * (-self.c_1_ / (sigmas + 1e-4) + self.c_2__)

  def get_diffusion_time(self, config, batch_size, batch_device, t_min, importance_sampling=False):
    time = torch.rand(batch_size, device=batch_device) * (1. / t_min - 1. / self.T) + 1. / self.T
    return 1. / time, 1

  def get_t_min(self, config, st=False):
    if st:
      max_ = np.random.rand() * (1. / self.eps - 1. / self.T) + 1. / self.T
      return 1. / max_
    return self.eps


def get_sde(config, state):
  """Factory keyed on ``config.training.sde`` (sde_lib.py:433-445)."""
  name = config.training.sde.lower()
  m = config.model
  if name == 'vpsde':
    return VPSDE(truncation_time=config.training.truncation_time, beta_min=m.beta_min,
                 beta_max=m.beta_max, N=m.num_scales)
  if name == 'subvpsde':
    return subVPSDE(truncation_time=config.training.truncation_time, beta_min=m.beta_min,
                    beta_max=m.beta_max, N=m.num_scales)
  if name == 'vesde':
    return VESDE(sigma_min=m.sigma_min, sigma_max=m.sigma_max, N=m.num_scales)
  if name == 'reciprocal_vesde':
    return reciprocal_VESDE(sigma_min=m.sigma_min, sigma_max=m.sigma_max, N=m.num_scales,
                            eta=config.training.eta)
  raise NotImplementedError(f"SDE {config.training.sde} unknown.")
