"""Training objective and the one-step training function (reference: losses.py).

Interface kept verbatim: ``get_optimizer(config, params)`` (:29-41), ``optimization_manager(config)``
-> ``optimize_fn(optimizer, params, step, lr, warmup, grad_clip)`` (:44-58),
``get_sde_loss_fn`` (:61-168), ``get_smld_loss_fn`` (:171-192), ``get_ddpm_loss_fn`` (:195-215),
``get_step_fn(config, sde, train, optimize_fn=None)`` -> ``step_fn(state, batch)`` returning the
per-sample losses as a CPU tensor and mutating ``state`` in place (:218-325).

What changes underneath: the score network is one planned HIP graph instead of
37.7k eager ops; ``optimizer`` is ``engine.optim.FusedAdam`` (clip + Adam in two launches, norm
kept on the device); the EMA update is one launch; and with ``torch.distributed`` initialised the
gradients are averaged by a bucketed RCCL all-reduce inside ``optimize_fn``.

The soft-truncation loss itself (time sampling, perturbation, weighting) is per-sample scalar
math plus a few element-wise passes over the [B,3,H,W] batch; it is written with the same torch
expression order as the reference so results are bit-identical on identical inputs.
"""
import numpy as np
import torch
import torch.optim as optim

from .engine import ddp
from .engine.flat import flat_of
from .engine.optim import FusedAdam
from .models import utils as mutils
from .sde_lib import VESDE, VPSDE


def get_optimizer(config, params):
  """Adam / AdamW per ``config.optim`` (losses.py:29-41).  Flat-backed parameters (the score
  network) get the fused HIP optimizer; other parameter lists get torch's."""
  params = list(params)
  o = config.optim
  fused = flat_of(params) is not None
  if o.optimizer == 'Adam':
    if fused:
      return FusedAdam(params, lr=o.lr, betas=(o.beta1, 0.999), eps=o.eps, weight_decay=o.weight_decay,
                       amsgrad=o.amsgrad)
    return optim.Adam(params, lr=o.lr, betas=(o.beta1, 0.999), eps=o.eps, weight_decay=o.weight_decay,
                      amsgrad=o.amsgrad)
  if o.optimizer == 'AdamW':
    if fused:
      return FusedAdam(params, lr=o.lr, betas=(o.beta1, 0.99), eps=o.eps, weight_decay=o.weight_decay,
                       adamw=True)
    return optim.AdamW(params, lr=o.lr, betas=(o.beta1, 0.99), eps=o.eps, weight_decay=o.weight_decay)
  raise NotImplementedError(f'Optimizer {o.optimizer} not supported yet!')


def optimization_manager(config):
  """optimize_fn: lr warm-up, gradient exchange (multi-GPU), clipping, optimizer step (losses.py:44-58)."""

  def optimize_fn(optimizer, params, step, lr=config.optim.lr, warmup=config.optim.warmup,
                  grad_clip=config.optim.grad_clip):
    params = list(params)
    if warmup > 0:
      for g in optimizer.param_groups:
        g['lr'] = lr * np.minimum(step / warmup, 1.0)
    ddp.sync_gradients(optimizer, None if hasattr(optimizer, 'clip_grad_norm') else params)
    if grad_clip >= 0:
      if hasattr(optimizer, 'clip_grad_norm'):
        optimizer.clip_grad_norm(grad_clip)
      else:
        torch.nn.utils.clip_grad_norm_(params, max_norm=grad_clip)
    optimizer.step()

  return optimize_fn


def get_sde_loss_fn(config, sde, train, variance='scoreflow'):
  """Soft-truncation weighted denoising score matching for a continuous SDE (losses.py:61-168)."""
  reduce_op = torch.mean if config.training.reduce_mean else lambda *args, **kwargs: 0.5 * torch.sum(*args, **kwargs)

  def approx_standard_normal_cdf(x):
    return 0.5 * (1.0 + torch.tanh(np.sqrt(2.0 / np.pi) * (x + 0.044715 * (x ** 3))))

  def discretized_gaussian_log_likelihood(x, means, log_scales):
    assert x.shape == means.shape
    centered_x = x - means
    inv_stdv = torch.exp(-log_scales)
    cdf_plus = approx_standard_normal_cdf(inv_stdv * (centered_x + 1. / 255.))
    cdf_min = approx_standard_normal_cdf(inv_stdv * (centered_x - 1. / 255.))
    floor = torch.tensor(1e-12, device=cdf_plus.device)
    log_cdf_plus = torch.log(torch.max(cdf_plus, floor))
    log_one_minus_cdf_min = torch.log(torch.max(1. - cdf_min, floor))
    cdf_delta = cdf_plus - cdf_min
    log_probs = torch.where(x < -0.999, log_cdf_plus,
                            torch.where(x > 0.999, log_one_minus_cdf_min,
                                        torch.log(torch.max(cdf_delta, floor))))
    assert log_probs.shape == x.shape
    return log_probs

  def loss_fn(model, batch, importance_sampling, t_min=None):
    """Per-sample losses [B] for one (micro-)batch (losses.py:101-166)."""
    if t_min is None:
      t_min = sde.get_t_min(config)
    t, Z = sde.get_diffusion_time(config, batch.shape[0], batch.device, t_min,
                                  importance_sampling=importance_sampling)
    score_fn = mutils.get_score_fn(config, sde, model, train=train, continuous=config.training.continuous)
    z = torch.randn_like(batch)
    mean, std = sde.marginal_prob(batch, t)
    perturbed_data = mean + std[:, None, None, None] * z
    score = score_fn(perturbed_data, t)

    if config.training.importance_sampling or not config.training.likelihood_weighting:
      losses = torch.square(score * std[:, None, None, None] + z)
      losses = 0.5 * Z * reduce_op(losses.reshape(losses.shape[0], -1), dim=-1)
    else:
      g2 = sde.sde(torch.zeros_like(batch), t)[1] ** 2
      losses = torch.square(score + z / std[:, None, None, None])
      losses = 0.5 * Z * reduce_op(losses.reshape(losses.shape[0], -1), dim=-1) * g2

    if config.training.reconstruction_loss:
      eps_vec = torch.ones((batch.shape[0]), device=batch.device) * t_min
      mean, std = sde.marginal_prob(batch, eps_vec)
      z = torch.randn_like(batch)
      perturbed_data = mean + std[:, None, None, None] * z
      score = score_fn(perturbed_data, eps_vec)
      alpha, beta = sde.marginal_prob(torch.ones_like(batch), eps_vec)
      q_mean = perturbed_data / alpha + beta[:, None, None, None] ** 2 * score / alpha
      if variance == 'ddpm':
        q_std = beta
      elif variance == 'scoreflow':
        q_std = beta / torch.mean(alpha, axis=(1, 2, 3))
      if config.data.dequantization == 'lossless':
        decoder_nll = -discretized_gaussian_log_likelihood(
          batch, means=q_mean, log_scales=torch.log(q_std)[:, None, None, None])
        reconstruction_loss = decoder_nll.sum(axis=(1, 2, 3))
      else:
        n_dim = np.prod(batch.shape[1:])
        p_entropy = n_dim / 2. * (np.log(2 * np.pi) + 2 * torch.log(std) + 1.)
        q_recon = n_dim / 2. * (np.log(2 * np.pi) + 2 * torch.log(q_std)) \
                  + 0.5 / (q_std ** 2) * torch.square(batch - q_mean).sum(axis=(1, 2, 3))
        assert q_recon.shape == p_entropy.shape == torch.Size([batch.shape[0]])
        reconstruction_loss = q_recon - p_entropy
        assert losses.shape == reconstruction_loss.shape
      if config.training.reduce_mean:
        reconstruction_loss = reconstruction_loss / np.prod(list(batch.shape[1:]))
      losses = losses + reconstruction_loss

    return losses

  return loss_fn


def get_smld_loss_fn(config, vesde, train):
  """Legacy discrete SMLD objective (losses.py:171-192); unreachable with continuous configs."""
  assert isinstance(vesde, VESDE), "SMLD training only works for VESDEs."
  smld_sigma_array = torch.flip(vesde.discrete_sigmas, dims=(0,))
  reduce_op = torch.mean if config.training.reduce_mean else lambda *args, **kwargs: 0.5 * torch.sum(*args, **kwargs)

  def loss_fn(model, batch):
    model_fn = mutils.get_model_fn(model, train=train)
    labels = torch.randint(0, vesde.N, (batch.shape[0],), device=batch.device)
    sigmas = smld_sigma_array.to(batch.device)[labels]
    noise = torch.randn_like(batch) * sigmas[:, None, None, None]
    perturbed_data = noise + batch
    score = model_fn(perturbed_data, labels)
    target = -noise / (sigmas ** 2)[:, None, None, None]
    losses = torch.square(score - target)
    losses = reduce_op(losses.reshape(losses.shape[0], -1), dim=-1) * sigmas ** 2
    return torch.mean(losses)

  return loss_fn


def get_ddpm_loss_fn(config, vpsde, train):
  """Legacy discrete DDPM objective (losses.py:195-215); unreachable with continuous configs."""
  assert isinstance(vpsde, VPSDE), "DDPM training only works for VPSDEs."
  reduce_op = torch.mean if config.training.reduce_mean else lambda *args, **kwargs: 0.5 * torch.sum(*args, **kwargs)

  def loss_fn(model, batch):
    model_fn = mutils.get_model_fn(model, train=train)
    labels = torch.randint(0, vpsde.N, (batch.shape[0],), device=batch.device)
    sqrt_alphas_cumprod = vpsde.sqrt_alphas_cumprod.to(batch.device)
    sqrt_1m_alphas_cumprod = vpsde.sqrt_1m_alphas_cumprod.to(batch.device)
    noise = torch.randn_like(batch)
    perturbed_data = sqrt_alphas_cumprod[labels, None, None, None] * batch + \
                     sqrt_1m_alphas_cumprod[labels, None, None, None] * noise
    score = model_fn(perturbed_data, labels)
    losses = torch.square(score - noise)
    losses = reduce_op(losses.reshape(losses.shape[0], -1), dim=-1)
    return torch.mean(losses)

  return loss_fn


def get_step_fn(config, sde, train, optimize_fn=None):
  """One training step: ``step_fn(state, batch) -> losses[B]`` on the CPU (losses.py:218-325).

  ``state`` = {'model', 'optimizer', 'ema', 'step'}; mutated in place exactly like the reference:
  zero_grad, one host draw of t_min shared by the micro-batches, per-micro-batch
  loss/backward, optimize_fn, ``step += 1``, EMA update.
  """
  if config.training.continuous:
    loss_fn = get_sde_loss_fn(config, sde, train)
  else:
    assert not config.training.likelihood_weighting, \
      "Likelihood weighting is not supported for original SMLD/DDPM training."
    if isinstance(sde, VESDE):
      loss_fn = get_smld_loss_fn(config, sde, train)
    elif isinstance(sde, VPSDE):
      loss_fn = get_ddpm_loss_fn(config, sde, train)
    else:
      raise ValueError(f"Discrete training for {sde.__class__.__name__} is not recommended.")

  def _finish(state, model, optimizer):
    optimize_fn(optimizer, model.parameters(), step=state['step'])
    state['step'] += 1
    state['ema'].update(model.parameters())

  def step_fn(state, batch):
    model = state['model']
    optimizer = state['optimizer']
    if train:
      optimizer.zero_grad()
      batch_size = batch.shape[0]
      nmb = config.optim.num_micro_batch
      per = batch_size // nmb
      losses_ = torch.zeros(batch_size)
      t_min = sde.get_t_min(config)
      for k in range(nmb):
        losses = loss_fn(model, batch[per * k: per * (k + 1)],
                         importance_sampling=config.training.importance_sampling, t_min=t_min)
        torch.mean(losses).backward(retain_graph=True)
        losses_[per * k: per * (k + 1)] = losses.cpu().detach()
      _finish(state, model, optimizer)
    return losses_

  def step_fn_mixed(state, batch):
    """Half of each micro-batch importance-sampled, half uniform-time (losses.py:295-320)."""
    model = state['model']
    optimizer = state['optimizer']
    if train:
      optimizer.zero_grad()
      batch_size = batch.shape[0]
      nmb = config.optim.num_micro_batch
      per = batch_size // nmb
      half = batch_size // (2 * nmb)
      losses_ = torch.zeros(batch_size // 2)
      t_min = sde.get_t_min(config)
      for k in range(nmb):
        losses_is = loss_fn(model, batch[per * k: per * k + half], importance_sampling=True, t_min=t_min)
        losses_ddpm = loss_fn(model, batch[per * k + half: per * (k + 1)], importance_sampling=False, t_min=t_min)
        if config.training.balanced:
          losses = losses_is + config.training.ddpm_weight * \
                   torch.mean(losses_is / losses_ddpm).detach().item() * losses_ddpm
        else:
          losses = losses_is + config.training.ddpm_weight * losses_ddpm
        torch.mean(losses).backward(retain_graph=True)
        losses_[per // 2 * k: per // 2 * (k + 1)] = losses.cpu().detach()
      _finish(state, model, optimizer)
    return losses_

  return step_fn_mixed if config.training.mixed else step_fn


def get_div_fn(fn):
  """Hutchinson-Skilling divergence estimator (losses.py:327-338)."""

  def div_fn(x, t, eps):
    with torch.enable_grad():
      fn_eps = torch.sum(fn(x, t) * eps)
      grad_fn_eps = torch.autograd.grad(fn_eps, x)[0]
    return torch.sum(grad_fn_eps * eps, dim=tuple(range(1, len(x.shape))))

  return div_fn
