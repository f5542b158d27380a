"""Likelihood evaluation on the score network: NLL by the probability-flow ODE and the soft-truncation NELBO.

Mirror of the reference's ``likelihood.py`` -- ``get_div_fn`` (:27-39), ``get_likelihood_fn`` (:42-134),
``get_elbo_fn`` (:136-208), ``get_likelihood_residual_fn`` (:210-313) -- with the same call signatures, return
values and noise-draw order, so ``utils.get_loss_fns`` (utils.py:78-79) gets working ``nll_fn`` / ``nelbo_fn``.
Everything here is arithmetic around ``score_fn`` and its input gradient: the network and its hand-written
backward (with respect to the input) run on the HIP engine; the ODE solver is SciPy's RK45 restated on device tensors
(engine/rk45.py), the [x, log p] state never leaves the device.
"""
import numpy as np
import torch
from scipy import integrate

from . import _decoder
from .engine import rk45
from .models import utils as mutils


def _noise_like(x, hutchinson_type):
  if hutchinson_type == 'Gaussian':
    return torch.randn_like(x)
  if hutchinson_type == 'Rademacher':
    return torch.randint_like(x, low=0, high=2).float() * 2 - 1.
  raise NotImplementedError(f'Hutchinson type {hutchinson_type} unknown.')


def _bcast(v):
  return v[:, None, None, None]


def get_div_fn(fn):
  """Hutchinson-Skilling estimate of div fn: ``eps^T (d fn / dx) eps`` per sample (likelihood.py:27-39)."""

  def div_fn(x, t, eps):
    with torch.enable_grad():
      x.requires_grad_(True)
      projected = torch.sum(fn(x, t) * eps)
      grad = torch.autograd.grad(projected, x)[0]
    x.requires_grad_(False)
    return torch.sum(grad * eps, dim=tuple(range(1, len(x.shape))))

  return div_fn


def get_likelihood_residual_fn(config, sde, score_fn, variance='ddpm'):
  """Reconstruction term at the truncation time (likelihood.py:210-313): ``fn(batch, eps=None) -> [B]`` nats,
  decoder negative log-likelihood minus the entropy of the perturbation kernel (`_decoder.py`)."""

  def residual_lossless(batch, eps=None):
    std, q_mean, q_std = _decoder.posterior_at(sde, score_fn, batch, sde.eps if eps is None else eps, variance)
    if not config.data.centered:
      batch, q_mean, q_std = 2. * batch - 1., 2. * q_mean - 1., 2. * q_std
    nll = -_decoder.discretized_gaussian_log_likelihood(batch, means=q_mean, log_scales=_bcast(torch.log(q_std)))
    residual = nll.sum(axis=(1, 2, 3)) - _decoder.entropy_of_perturbation(np.prod(batch.shape[1:]), std)
    assert residual.shape == torch.Size([batch.shape[0]])
    return residual

  def residual_gaussian(batch, eps=None):
    std, q_mean, q_std = _decoder.posterior_at(sde, score_fn, batch, sde.eps if eps is None else eps, variance)
    p_entropy = _decoder.entropy_of_perturbation(np.prod(batch.shape[1:]), std)
    q_recon = _decoder.gaussian_reconstruction(batch, q_mean, q_std)
    assert q_recon.shape == p_entropy.shape == torch.Size([batch.shape[0]])
    return q_recon - p_entropy

  return residual_lossless if config.data.dequantization == 'lossless' else residual_gaussian


def get_likelihood_fn(config, sde, inverse_scaler, hutchinson_type='Rademacher', rtol=1e-5, atol=1e-5, method='RK45'):
  """``likelihood_fn(model, data, logdet=0., eps=1e-5, mode='correct') -> (bpd[B], z, nfe)`` (likelihood.py:42-134)."""

  def drift_fn(model, x, t):
    score_fn = mutils.get_score_fn(config, sde, model, train=False, continuous=True)
    rsde = sde.reverse(score_fn, probability_flow=config.eval.probability_flow, lambda_=config.eval.lambda_)
    return rsde.sde(x, t)[0]

  def div_fn(model, x, t, noise):
    return get_div_fn(lambda xx, tt: drift_fn(model, xx, tt))(x, t, noise)

  def likelihood_fn(model, data, logdet=0., eps=1e-5, mode='correct'):
    # hundreds of network evaluations on fixed parameters: prepare the convolution weights once
    with torch.no_grad(), mutils.frozen_weights(model):
      score_fn = mutils.get_score_fn(config, sde, model, train=False, continuous=True)
      shape = data.shape
      B = shape[0]
      epsilon = _noise_like(data, hutchinson_type)
      if mode == 'correct':
        z = torch.randn_like(data)
        mean, std = sde.marginal_prob(data, torch.ones(B, device=data.device) * eps)
        start = mean + _bcast(std) * z
      elif mode == 'wrong':
        start = data
      else:
        raise NotImplementedError

      # augmented state [x (flattened), delta log p (B)], float64 as in the reference's numpy state
      def rhs(t, sample):
        vec_t = torch.ones(B, device=sample.device) * t
        return drift_fn(model, sample, vec_t), div_fn(model, sample, vec_t, epsilon)

      if method == 'RK45':
        def ode_func(t, y):
          sample = y[:-B].reshape(shape).to(torch.float32)
          drift, logp_grad = rhs(t, sample)
          return torch.cat([drift.reshape(-1).to(torch.float64), logp_grad.reshape(-1).to(torch.float64)])

        init = torch.cat([start.reshape(-1).to(torch.float64), torch.zeros(B, dtype=torch.float64, device=data.device)])
        zp, nfe = rk45.solve_ivp_rk45(ode_func, (eps, sde.T), init, rtol=rtol, atol=atol)
      else:
        def ode_func(t, y):
          sample = mutils.from_flattened_numpy(y[:-B], shape).to(data.device).type(torch.float32)
          drift, logp_grad = rhs(t, sample)
          return np.concatenate([mutils.to_flattened_numpy(drift), mutils.to_flattened_numpy(logp_grad)], axis=0)

        init = np.concatenate([mutils.to_flattened_numpy(start), np.zeros((B,))], axis=0)
        solution = integrate.solve_ivp(ode_func, (eps, sde.T), init, rtol=rtol, atol=atol, method=method)
        nfe = solution.nfev
        zp = torch.from_numpy(solution.y[:, -1]).to(data.device)
      z = zp[:-B].reshape(shape).to(data.device).type(torch.float32)
      delta_logp = zp[-B:].reshape((B,)).to(data.device).type(torch.float32)
      prior_logp = sde.prior_logp(z)
      if mode == 'correct':
        residual_fn = get_likelihood_residual_fn(config, sde, score_fn, variance='scoreflow')
        delta_logp = delta_logp - residual_fn(data, eps)
      bpd = -(prior_logp + delta_logp + logdet) / np.log(2)
      bpd = bpd / np.prod(shape[1:])
      bpd = bpd + (7. - inverse_scaler(-1.))          # log-likelihood -> bits/dim of the 8-bit data
      return bpd, z, nfe

  return likelihood_fn


def get_elbo_fn(config, sde, inverse_scaler=None, hutchinson_type='Rademacher'):
  """Importance-sampled soft-truncation NELBO: ``loss_fn(model, batch, logdet=0., eps=1e-5) -> (nelbo_bpd[B],
  residual_bpd[B])`` (likelihood.py:136-208)."""

  @torch.enable_grad()
  def loss_fn(model, batch, logdet=0., eps=1e-5):
    score_fn = mutils.get_score_fn(config, sde, model, train=False, continuous=True)
    B = batch.shape[0]
    rve = config.training.sde.lower() == 'reciprocal_vesde'
    time, Z = sde.get_diffusion_time(config, B, batch.device, eps, importance_sampling=True)
    qt = 1. / (1. / eps - 1. / sde.T) if rve else 1 / (sde.T - eps)
    z = torch.randn_like(batch)
    mean, std = sde.marginal_prob(batch, time)
    perturbed = (mean + _bcast(std) * z).requires_grad_()
    score = score_fn(perturbed, time)
    f, g = sde.sde(perturbed, time)
    s = _bcast(std)
    a = s * score
    mu = (s ** 2) * score - (s ** 2) / (_bcast(g) ** 2) * f
    epsilon = _noise_like(batch, hutchinson_type)
    jvp = torch.autograd.grad(mu, perturbed, epsilon, create_graph=False)[0]
    Mu = -(jvp * epsilon).reshape(B, -1).sum(1, keepdim=False) * Z / qt
    Nu = -(a ** 2).reshape(B, -1).sum(1, keepdim=False) * Z / 2 / qt

    lp_t = torch.ones_like(time) * sde.T
    lp_z = torch.randn_like(batch)
    lp_mean, lp_std = sde.marginal_prob(batch, lp_t)
    lp = sde.prior_logp(lp_mean + _bcast(lp_std) * lp_z)

    scale = 2. * eps * np.log(sde.sigma_max / sde.sigma_min) if rve else 1.
    elbos = lp + (Mu + Nu) * scale
    n_dim = np.prod(list(batch.shape[1:]))
    residual_fn = get_likelihood_residual_fn(config, sde, score_fn, variance='scoreflow')
    return (-(elbos + logdet) / n_dim / np.log(2) + 7. - inverse_scaler(-1.),
            residual_fn(batch, eps) / n_dim / np.log(2))

  return loss_fn
