"""The Gaussian decoder at the truncation time, shared by the training loss and the likelihood evaluators.

The reference writes this block out three times (losses.py:79-99,134-164; likelihood.py:214-313); here it lives once.
Given a clean batch x it draws x_eps ~ p(x_eps | x), evaluates the score there and forms the Gaussian
q(x | x_eps) = N(q_mean, q_std^2) implied by Tweedie's formula; the callers turn that into a reconstruction term
(continuous Gaussian or 8-bit discretised).  The order of the torch expressions -- and of the noise draw -- is the
reference's, so results agree bit for bit on identical inputs.
"""
import numpy as np
import torch


def approx_standard_normal_cdf(x):
  return 0.5 * (1.0 + torch.tanh(np.sqrt(2.0 / np.pi) * (x + 0.044715 * (x ** 3))))


def discretized_gaussian_log_likelihood(x, means, log_scales):
  """log-probability of 8-bit data rescaled to [-1, 1] under a Gaussian binned at 1/255 half-widths, with open bins
  at both ends (losses.py:82-99)."""
  assert x.shape == means.shape
  centred = x - means
  inv_std = torch.exp(-log_scales)
  upper = approx_standard_normal_cdf(inv_std * (centred + 1. / 255.))
  lower = approx_standard_normal_cdf(inv_std * (centred - 1. / 255.))
  floor = torch.tensor(1e-12, device=upper.device)
  log_upper = torch.log(torch.max(upper, floor))
  log_above = torch.log(torch.max(1. - lower, floor))
  log_bin = torch.log(torch.max(upper - lower, floor))
  out = torch.where(x < -0.999, log_upper, torch.where(x > 0.999, log_above, log_bin))
  assert out.shape == x.shape
  return out


def entropy_of_perturbation(n_dim, std):
  """Differential entropy of N(., std^2 I) in n_dim dimensions, per sample."""
  return n_dim / 2. * (np.log(2 * np.pi) + 2 * torch.log(std) + 1.)


def gaussian_reconstruction(batch, q_mean, q_std):
  """-log N(batch; q_mean, q_std^2 I) per sample."""
  n_dim = np.prod(batch.shape[1:])
  return n_dim / 2. * (np.log(2 * np.pi) + 2 * torch.log(q_std)) \
      + 0.5 / (q_std ** 2) * torch.square(batch - q_mean).sum(axis=(1, 2, 3))


def posterior_at(sde, score_fn, batch, eps, variance):
  """Draw x_eps, evaluate the score, return (std of the perturbation, q_mean, q_std).

  ``variance``: 'ddpm' -> q_std = beta;  'scoreflow' -> beta / mean(alpha)  (alpha, beta = marginal_prob(1, eps))."""
  eps_vec = torch.ones((batch.shape[0]), device=batch.device) * eps
  mean, std = sde.marginal_prob(batch, eps_vec)
  z = torch.randn_like(batch)
  perturbed = mean + std[:, None, None, None] * z
  score = score_fn(perturbed, eps_vec)
  alpha, beta = sde.marginal_prob(torch.ones_like(batch), eps_vec)
  q_mean = perturbed / alpha + beta[:, None, None, None] ** 2 * score / alpha
  if variance == 'ddpm':
    q_std = beta
  elif variance == 'scoreflow':
    q_std = beta / torch.mean(alpha, axis=(1, 2, 3))
  return std, q_mean, q_std
