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
import os

import numpy as np
import torch
import torch.optim as optim

from . import _decoder
from .engine import ddp
from .engine.flat import flat_of
from .engine.optim import FusedAdam
from .models import utils as mutils
from .sde_lib import VESDE, VPSDE, reciprocal_VESDE


def _wide(v):
  return v[:, None, None, None]


def _per_sample_reducer(config):
  """mean over CHW when ``training.reduce_mean`` else half the sum (losses.py:77)."""
  if config.training.reduce_mean:
    return torch.mean
  return lambda *args, **kwargs: 0.5 * torch.sum(*args, **kwargs)


# optimizer name -> (second Adam beta, decoupled weight decay); losses.py:31-39
_OPTIMIZERS = {'Adam': (0.999, False), 'AdamW': (0.99, True)}


def get_optimizer(config, params):
  """Adam / AdamW per ``config.optim`` (losses.py:29-41).  Flat-backed parameters (the score network) get the fused
  HIP optimizer; other parameter lists get torch's."""
  params, o = list(params), config.optim
  if o.optimizer not in _OPTIMIZERS:
    raise NotImplementedError(f'Optimizer {o.optimizer} not supported yet!')
  beta2, decoupled = _OPTIMIZERS[o.optimizer]
  common = dict(lr=o.lr, betas=(o.beta1, beta2), eps=o.eps, weight_decay=o.weight_decay)
  if flat_of(params) is not None:
    return FusedAdam(params, adamw=True, **common) if decoupled else FusedAdam(params, amsgrad=o.amsgrad, **common)
  return optim.AdamW(params, **common) if decoupled else optim.Adam(params, amsgrad=o.amsgrad, **common)


def optimization_manager(config):
  """optimize_fn: lr warm-up, gradient exchange (multi-GPU), clipping, optimizer step (losses.py:44-58)."""

  def optimize_fn(optimizer, params, step, lr=config.optim.lr, warmup=config.optim.warmup,
                  grad_clip=config.optim.grad_clip):
    fused = hasattr(optimizer, 'clip_grad_norm')
    # a generator would be exhausted by the first consumer below; the fused optimizer works on its flat buffers and needs
    # no list at all (walking the module tree of the 61.8 M-parameter net is ~0.9 ms of host time)
    params = None if fused else list(params)
    if warmup > 0:
      warm_lr = lr * np.minimum(step / warmup, 1.0)
      for group in optimizer.param_groups:
        group['lr'] = warm_lr
    ddp.sync_gradients(optimizer, None if fused else params)
    if grad_clip >= 0:
      if fused:
        optimizer.clip_grad_norm(grad_clip)      # norm stays on the device, folded into the Adam launch
      else:
        torch.nn.utils.clip_grad_norm_(params, max_norm=grad_clip)
    optimizer.step()

  return optimize_fn


# STK_FUSED_LOSS=0: the loss arithmetic around the network as the reference's torch expressions (~35 small launches per
# evaluation and as many in its backward) instead of three kernels (A/B switch; results agree to the last bits)
FUSED_LOSS = os.environ.get('STK_FUSED_LOSS', '1') != '0'


class _ScoreMatching(torch.autograd.Function):
  """losses[n] = wgt[n] * reduce((score * std + z)^2) with score = -net / std (VP) or net -- the tail of the reference's
  loss_fn (losses.py:122-132 behind models/utils.py:160) as ONE kernel per direction (stk_sm_loss_*) on the raw network
  output, every element-wise operation rounded as torch rounds it.

  The forward kernel gives a sample to one workgroup: fine for 128 x 3072 (CIFAR-10), 614 us for 4 x 196608 (256 x 256 at
  batch 4).  Large samples are therefore cut into `S` equal pieces that the kernel sees as samples of their own (half the
  sum of squares each, `reduce_mean` off); the pieces of a sample are added here in a fixed order."""

  PIECE = 16384          # elements per workgroup when a sample is cut up

  @staticmethod
  def _pieces(B, inner):
    S = 1
    while inner // S > _ScoreMatching.PIECE and (inner // S) % 8 == 0 and B * S < 4096:
      S *= 2
    return S

  @staticmethod
  def forward(ctx, net, z, std, wgt, lib, neg_over_std, reduce_mean):
    from .engine import lib as stk_lib
    net, z, std, wgt = net.contiguous(), z.contiguous(), std.contiguous(), wgt.contiguous()
    B, inner = net.shape[0], net[0].numel()
    S = _ScoreMatching._pieces(B, inner)
    stdS = std.repeat_interleave(S) if S > 1 else std
    wgtS = wgt.repeat_interleave(S) if S > 1 else wgt
    rm = int(reduce_mean) if S == 1 else 0
    part = torch.empty(B * S, dtype=torch.float32, device=net.device)
    with stk_lib.device_guard(net.device):
      lib.sm_loss_fwd_f32(net.data_ptr(), z.data_ptr(), stdS.data_ptr(), wgtS.data_ptr(), part.data_ptr(), B * S, inner // S,
                          int(neg_over_std), 0, rm, stk_lib.stream_ptr(net.device))
    if S > 1:
      losses = part.view(B, S).sum(dim=1)                 # = wgt * 0.5 * sum r^2
      if reduce_mean:
        losses = losses * (2.0 / inner)
    else:
      losses = part
    ctx.save_for_backward(net, z, stdS, wgtS)
    ctx.args = (lib, B, inner, S, int(neg_over_std), int(reduce_mean))
    return losses

  @staticmethod
  def backward(ctx, dloss):
    from .engine import lib as stk_lib
    net, z, stdS, wgtS = ctx.saved_tensors
    lib, B, inner, S, vp, reduce_mean = ctx.args
    dnet = torch.empty_like(net)
    dloss = dloss.contiguous()
    rm = reduce_mean
    if S > 1:
      dloss = (dloss * (2.0 / inner) if reduce_mean else dloss).repeat_interleave(S)
      rm = 0
    with stk_lib.device_guard(net.device):
      lib.sm_loss_bwd_f32(net.data_ptr(), z.data_ptr(), stdS.data_ptr(), wgtS.data_ptr(), dloss.data_ptr(), dnet.data_ptr(),
                          B * S, inner // S, vp, 0, rm, stk_lib.stream_ptr(net.device))
    return dnet, None, None, None, None, None, None


_fused_bufs = {}


def _engine_lib(model, batch):
  """The kernel library of a score network that runs on the engine, when `batch` lives where that library computes."""
  net = getattr(model, 'module', model)
  engine = getattr(net, 'engine', None)
  if engine is None:
    return None
  lib = engine().lib
  if lib.is_device != batch.is_cuda or batch.dtype != torch.float32 or not hasattr(lib, 'sm_loss_fwd_f32'):
    return None
  return lib


def get_sde_loss_fn(config, sde, train, variance='scoreflow'):
  """Soft-truncation weighted denoising score matching for a continuous SDE (losses.py:61-168):
  ``loss_fn(model, batch, importance_sampling, t_min=None) -> [B]``."""
  tr = config.training
  reduce_op = _per_sample_reducer(config)

  def score_matching(batch, t, Z, z, std, score):
    if tr.importance_sampling or not tr.likelihood_weighting:
      sq = torch.square(score * _wide(std) + z)
      return 0.5 * Z * reduce_op(sq.reshape(sq.shape[0], -1), dim=-1)
    g2 = sde.sde(torch.zeros_like(batch), t)[1] ** 2
    sq = torch.square(score + z / _wide(std))
    return 0.5 * Z * reduce_op(sq.reshape(sq.shape[0], -1), dim=-1) * g2

  def reconstruction(score_fn, batch, t_min, losses):
    """Decoder term at the truncation time (losses.py:134-164; off in every BASELINE config)."""
    std, q_mean, q_std = _decoder.posterior_at(sde, score_fn, batch, t_min, variance)
    if config.data.dequantization == 'lossless':
      nll = -_decoder.discretized_gaussian_log_likelihood(batch, means=q_mean, log_scales=_wide(torch.log(q_std)))
      term = nll.sum(axis=(1, 2, 3))
    else:
      p_entropy = _decoder.entropy_of_perturbation(np.prod(batch.shape[1:]), std)
      q_recon = _decoder.gaussian_reconstruction(batch, q_mean, q_std)
      assert q_recon.shape == p_entropy.shape == torch.Size([batch.shape[0]])
      term = q_recon - p_entropy
      assert losses.shape == term.shape
    return term / np.prod(list(batch.shape[1:])) if tr.reduce_mean else term

  def fused_losses(lib, raw_fn, neg_over_std, batch, t, Z):
    """The same values from three kernels: x_t = mean + std z (stk_perturb_f32, bit-identical to the torch expression),
    the network, the weighted squared residual per sample (stk_sm_loss_fwd_f32 / _bwd_f32)."""
    from .engine import lib as stk_lib
    B = batch.shape[0]
    z = torch.randn_like(batch)
    key = (B, batch.dtype, batch.device)
    bufs = _fused_bufs.get(key)
    if bufs is None:       # per batch size: the [B,1,1,1] ones the coefficients are read off (never written again)
      bufs = _fused_bufs[key] = torch.ones((B, 1, 1, 1), dtype=batch.dtype, device=batch.device)
    ones = bufs
    coeff, std = sde.marginal_prob(ones, t)                    # mean = coeff * x: [B,1,1,1] coefficients, [B] std
    std = std.to(torch.float32).contiguous()
    x = batch.contiguous()
    xt = torch.empty_like(x)
    a = None if isinstance(sde, (VESDE, reciprocal_VESDE)) else coeff.reshape(B).to(torch.float32).contiguous()
    with stk_lib.device_guard(x.device):
      lib.perturb_f32(x.data_ptr(), z.data_ptr(), a.data_ptr() if a is not None else None, std.data_ptr(), xt.data_ptr(),
                      B, x[0].numel(), stk_lib.stream_ptr(x.device))
    net = raw_fn(xt, t)
    if torch.is_tensor(Z) and Z.device.type != 'cpu':
      wgt = (0.5 * Z) * torch.ones(B, dtype=torch.float32, device=x.device)
    else:                  # a host scalar (the normalising constant of the importance-sampled times): a fill, no host-to-device
      # copy.  (A fresh tensor per evaluation: the autograd node keeps it, and `training.mixed` holds two evaluations at once.)
      wgt = torch.full((B,), 0.5 * float(Z), dtype=torch.float32, device=x.device)
    return _ScoreMatching.apply(net, z, std, wgt, lib, neg_over_std, bool(tr.reduce_mean))

  def loss_fn(model, batch, importance_sampling, t_min=None):
    if t_min is None:
      t_min = sde.get_t_min(config)
    t, Z = sde.get_diffusion_time(config, batch.shape[0], batch.device, t_min, importance_sampling=importance_sampling)
    if FUSED_LOSS and not tr.reconstruction_loss and (tr.importance_sampling or not tr.likelihood_weighting):
      lib = _engine_lib(model, batch)
      raw_fn, neg_over_std = mutils.get_raw_fn(config, sde, model, train=train, continuous=tr.continuous) if lib is not None else (None, None)
      if raw_fn is not None:
        return fused_losses(lib, raw_fn, neg_over_std, batch, t, Z)
    score_fn = mutils.get_score_fn(config, sde, model, train=train, continuous=tr.continuous)
    z = torch.randn_like(batch)
    mean, std = sde.marginal_prob(batch, t)
    score = score_fn(mean + _wide(std) * z, t)
    losses = score_matching(batch, t, Z, z, std, score)
    if tr.reconstruction_loss:
      losses = losses + reconstruction(score_fn, batch, t_min, losses)
    return losses

  return loss_fn


def get_smld_loss_fn(config, vesde, train):
  """Legacy discrete SMLD objective (losses.py:171-192); unreachable with continuous configs."""
  assert isinstance(vesde, VESDE), "SMLD training only works for VESDEs."
  descending = torch.flip(vesde.discrete_sigmas, dims=(0,))
  reduce_op = _per_sample_reducer(config)

  def loss_fn(model, batch):
    model_fn = mutils.get_model_fn(model, train=train)
    labels = torch.randint(0, vesde.N, (batch.shape[0],), device=batch.device)
    sigmas = descending.to(batch.device)[labels]
    noise = torch.randn_like(batch) * _wide(sigmas)
    score = model_fn(noise + batch, labels)
    sq = torch.square(score - (-noise / _wide(sigmas ** 2)))
    return torch.mean(reduce_op(sq.reshape(sq.shape[0], -1), dim=-1) * sigmas ** 2)

  return loss_fn


def get_ddpm_loss_fn(config, vpsde, train):
  """Legacy discrete DDPM objective (losses.py:195-215); unreachable with continuous configs."""
  assert isinstance(vpsde, VPSDE), "DDPM training only works for VPSDEs."
  reduce_op = _per_sample_reducer(config)

  def loss_fn(model, batch):
    model_fn = mutils.get_model_fn(model, train=train)
    labels = torch.randint(0, vpsde.N, (batch.shape[0],), device=batch.device)
    keep = vpsde.sqrt_alphas_cumprod.to(batch.device)
    blur = vpsde.sqrt_1m_alphas_cumprod.to(batch.device)
    noise = torch.randn_like(batch)
    noisy = keep[labels, None, None, None] * batch + blur[labels, None, None, None] * noise
    sq = torch.square(model_fn(noisy, labels) - noise)
    return torch.mean(reduce_op(sq.reshape(sq.shape[0], -1), dim=-1))

  return loss_fn


def _pick_loss_fn(config, sde, train):
  if config.training.continuous:
    return get_sde_loss_fn(config, sde, train)
  assert not config.training.likelihood_weighting, \
    "Likelihood weighting is not supported for original SMLD/DDPM training."
  if isinstance(sde, VESDE):
    return get_smld_loss_fn(config, sde, train)
  if isinstance(sde, VPSDE):
    return get_ddpm_loss_fn(config, sde, train)
  raise ValueError(f"Discrete training for {sde.__class__.__name__} is not recommended.")


# STK_DDP_OVERLAP=0: exchange the gradients after the backward (one bucketed all-reduce) instead of during it
OVERLAP_EXCHANGE = os.environ.get('STK_DDP_OVERLAP', '1') != '0'
# STK_RANGE_CHECK=K: every K-th training step (and the first) the per-image maxima of the fp32 output gradients are computed
# (Executor.dynamic_range_report: one device reduction per 3x3 layer, one transfer) and a warning is issued when an image lies
# more than RANGE_DECADES below the batch maximum of some layer -- beyond that the one-scale-per-tensor split convolutions no
# longer give that image fp32 accuracy.  Unset: every 1000th step for likelihood-weighted losses (g^2 weights spread the
# per-sample gradients over decades, reference losses.py:126-129), never otherwise.  0 = off.  With training.mixed the report
# covers the last network evaluation of the step.
def _env_int(name):
  v = os.environ.get(name)
  if v is None or v.strip() == '':
    return None
  try:
    return int(v)
  except ValueError:
    import warnings
    warnings.warn(f'{name}={v!r} is not an integer: ignored')
    return None


RANGE_CHECK_EVERY = _env_int('STK_RANGE_CHECK')
RANGE_DECADES = 5.0
# STK_ASYNC_LOSS=0: fetch the per-sample losses with a blocking .cpu() after the backward, as the reference does (A/B switch)
ASYNC_LOSS_COPY = os.environ.get('STK_ASYNC_LOSS', '1') != '0'


def _warn_dynamic_range(model, step):
  """See RANGE_CHECK_EVERY."""
  net = getattr(model, 'module', model)
  engine = getattr(net, 'engine', None)
  if engine is None:
    return None
  rows = engine().dynamic_range_report()
  if rows and rows[0][3] > RANGE_DECADES:
    import warnings
    name, hi, lo, dec = rows[0]
    warnings.warn(f'step {step}: the output gradients of {sum(r[3] > RANGE_DECADES for r in rows)} convolution(s) span more than '
                  f'{RANGE_DECADES:g} decades across the images of the batch (worst: {name}, image maxima {lo:.3e} ... {hi:.3e}, '
                  f'{dec:.1f} decades): the split convolutions scale a tensor by one power of two, so the gradient contribution of '
                  f'the faintest images is no longer fp32-accurate (DESIGN.md section 3); a smaller spread of the per-sample loss '
                  f'weights, or STK_PLANES=0 for an exact-fp32 check, tells whether it matters')
  return rows


def get_step_fn(config, sde, train, optimize_fn=None):
  """One training step: ``step_fn(state, batch) -> losses`` on the CPU (losses.py:218-325).

  ``state`` = {'model', 'optimizer', 'ema', 'step'}, mutated in place exactly like the reference: zero_grad, one host
  draw of t_min shared by the micro-batches, per-micro-batch loss + backward of its mean, ``optimize_fn``,
  ``step += 1``, EMA update.  With ``training.mixed`` each micro-batch is split in halves -- importance-sampled and
  uniform-time -- combined as L_is + w L_ddpm (optionally balanced by mean(L_is / L_ddpm)), and the function returns
  B/2 losses (losses.py:295-320)."""
  loss_fn = _pick_loss_fn(config, sde, train)
  tr = config.training
  mixed = bool(tr.mixed)

  def plain_losses(model, chunk, t_min):
    return loss_fn(model, chunk, importance_sampling=tr.importance_sampling, t_min=t_min)

  def mixed_losses(model, chunk, t_min):
    half = chunk.shape[0] // 2
    l_is = loss_fn(model, chunk[:half], importance_sampling=True, t_min=t_min)
    l_ddpm = loss_fn(model, chunk[half:], importance_sampling=False, t_min=t_min)
    weight = tr.ddpm_weight
    if tr.balanced:
      weight = weight * torch.mean(l_is / l_ddpm).detach().item()
    return l_is + weight * l_ddpm

  micro_losses = mixed_losses if mixed else plain_losses
  staging = {}       # number of losses -> pinned host buffer of the asynchronous device-to-host copy
  range_every = RANGE_CHECK_EVERY if RANGE_CHECK_EVERY is not None else (1000 if getattr(tr, 'likelihood_weighting', False) else 0)

  def step_fn(state, batch):
    model, optimizer = state['model'], state['optimizer']
    if not train:
      # the reference has no evaluation branch either: its `return losses_` raises exactly this (losses.py:279-293)
      raise UnboundLocalError("local variable 'losses_' referenced before assignment")
    optimizer.zero_grad()
    n, parts = batch.shape[0], config.optim.num_micro_batch
    per = n // parts
    out_per = per // 2 if mixed else per
    losses_ = torch.zeros(n // 2 if mixed else n)
    t_min = sde.get_t_min(config)
    pinned, copied = None, None
    ddp.begin_step(model)
    try:
      for k in range(parts):
        losses = micro_losses(model, batch[per * k: per * (k + 1)], t_min)
        if k == parts - 1 and OVERLAP_EXCHANGE:
          # multi-GPU: buckets of the flat gradient buffer are all-reduced as the last backward finishes them (with
          # several network evaluations per loss -- training.mixed -- as the LAST of their backwards does)
          ddp.arm_overlap(model)
        if losses.is_cuda and ASYNC_LOSS_COPY:
          # The reference's `losses.cpu()` (losses.py:288) after the backward makes the host wait for the whole backward
          # and only then launch clip / Adam / EMA and the next step: ~1.5-3 ms of idle GPU per 43 ms step.  The values
          # exist before the backward starts, so copy them to pinned memory asynchronously NOW (in stream order: after
          # the loss kernels, before the backward) and wait for that copy -- not for the backward -- when returning.
          if pinned is None:
            key = (losses_.numel(), losses.device)
            pinned = staging.get(key)
            if pinned is None:
              pinned = staging[key] = torch.empty(losses_.numel(), dtype=torch.float32).pin_memory()
            copied = torch.cuda.Event()
          pinned[out_per * k: out_per * (k + 1)].copy_(losses.detach(), non_blocking=True)
          copied.record(torch.cuda.current_stream(losses.device))     # the stream the copy was issued on
          torch.mean(losses).backward(retain_graph=True)
        else:
          torch.mean(losses).backward(retain_graph=True)
          losses_[out_per * k: out_per * (k + 1)] = losses.cpu().detach()
      optimize_fn(optimizer, model.parameters(), step=state['step'])
    except BaseException:
      ddp.disarm_overlap(model, wait=False)      # a step that raised must not leave its hook armed -- nor wait for its peers
      raise
    ddp.disarm_overlap(model)        # no-op after optimize_fn
    if range_every > 0 and state['step'] % range_every == 0:
      _warn_dynamic_range(model, state['step'])
    state['step'] += 1
    state['ema'].update(model.parameters())
    if pinned is not None:
      copied.synchronize()
      losses_.copy_(pinned)
    return losses_

  return step_fn


def get_div_fn(fn):
  """Hutchinson-Skilling divergence estimator (losses.py:327-338)."""

  def div_fn(x, t, eps):
    with torch.enable_grad():
      projected = torch.sum(fn(x, t) * eps)
      grad = torch.autograd.grad(projected, x)[0]
    return torch.sum(grad * eps, dim=tuple(range(1, len(x.shape))))

  return div_fn
