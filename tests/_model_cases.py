"""Model-level parity cases, shared by the CPU host-logic tests (checker backend) and the GPU parity
tests (HIP backend).  Every case compares the product path -- NCSNpp.forward on the planned-graph
engine, FusedAdam, the fused EMA, the samplers -- with the oracle RefNet driven by the same host code
and torch's own Adam, on identical inputs and explicitly injected noise.

Tolerance: the north star asks for loss/score tensors within 1e-3 relative (fp32) of the reference's
CPU path; the kernels are exact-fp32 MFMA, so the tests hold a tighter 2e-4.
"""
import copy

import numpy as np
import pytest
import torch

from _model_util import build_pair, grads_close, make_state, patched_rng, rel_err, tiny_config

TOL = 2e-4


def _inputs(cfg, sde, B, seed=1):
  g = torch.Generator().manual_seed(seed)
  H = cfg.data.image_size
  x = torch.randn(B, cfg.data.num_channels, H, H, generator=g)
  t = torch.rand(B, generator=g) * 0.9 + 0.05
  if cfg.model.embedding_type == 'positional':
    cond = t * 999
  else:
    cond = sde.marginal_prob(x, t)[1]
  return x, t, cond


def forward_backward(st, lib, family, B=3):
  cfg, cfg_cpu, sde, model, ref = build_pair(st, tiny_config(st, family), lib)
  dev = cfg.device
  x, t, cond = _inputs(cfg, sde, B)
  model.eval(); ref.eval()
  xg = x.clone().to(dev).requires_grad_(True)
  y = model(xg, cond.to(dev))
  xr = x.clone().requires_grad_(True)
  yr = ref(xr, cond)
  assert rel_err(y, yr) <= TOL, f'forward mismatch {rel_err(y, yr):.3e}'
  go = torch.randn(yr.shape, generator=torch.Generator().manual_seed(5))
  (y * go.to(dev)).sum().backward()
  (yr * go).sum().backward()
  assert rel_err(xg.grad, xr.grad) <= TOL, f'input-gradient mismatch {rel_err(xg.grad, xr.grad):.3e}'
  grads_close(model, ref, TOL)
  # a second forward under no_grad (sampling path) gives the same values
  with torch.no_grad():
    y2 = model(x.to(dev), cond.to(dev))
  assert torch.equal(y2, y.detach())


def score_fn_parity(st, lib, family):
  cfg, cfg_cpu, sde, model, ref = build_pair(st, tiny_config(st, family), lib)
  dev = cfg.device
  x, t, _ = _inputs(cfg, sde, 2)
  s = st.models.utils.get_score_fn(cfg, sde, model, train=False, continuous=True)(x.to(dev), t.to(dev))
  sr = st.models.utils.get_score_fn(cfg_cpu, sde, ref, train=False, continuous=True)(x, t)
  assert rel_err(s, sr) <= TOL


def train_steps(st, lib, family, steps=3, num_micro_batch=1, mixed=False, B=4, amsgrad=False):
  base = tiny_config(st, family)
  base.optim.amsgrad = bool(amsgrad)          # torch.optim.Adam(amsgrad=True) on the oracle side, stk_adam_amsgrad_f32 here
  base.optim.num_micro_batch = num_micro_batch
  base.optim.warmup = 2
  base.training.mixed = mixed
  cfg, cfg_cpu, sde, model, ref = build_pair(st, base, lib)
  dev = cfg.device
  state = make_state(st, cfg, model)
  assert type(state['optimizer']).__name__ == 'FusedAdam'
  state['optimizer']._backend = lib
  state['ema'].set_backend(lib)
  rstate = make_state(st, cfg_cpu, ref)
  assert type(rstate['optimizer']).__name__ == 'Adam'
  step_fn = st.losses.get_step_fn(cfg, sde, train=True, optimize_fn=st.losses.optimization_manager(cfg))
  rstep_fn = st.losses.get_step_fn(cfg_cpu, sde, train=True, optimize_fn=st.losses.optimization_manager(cfg_cpu))
  for i in range(steps):
    batch = st.datasets.synthetic_batch(cfg_cpu, B, generator=torch.Generator().manual_seed(100 + i))
    np.random.seed(7 + i)
    with patched_rng(50 + i):
      loss = step_fn(state, batch.to(dev))
    np.random.seed(7 + i)
    with patched_rng(50 + i):
      rloss = rstep_fn(rstate, batch)
    assert loss.device.type == 'cpu' and loss.shape == rloss.shape
    assert rel_err(loss, rloss) <= TOL, f'step {i}: loss mismatch {rel_err(loss, rloss):.3e}'
  assert state['step'] == rstate['step'] == steps
  # parameters after `steps` Adam updates.  Adam normalises the gradient (m/sqrt(v)), which amplifies
  # round-off in entries whose gradient is ~0, so parameters are compared in units of the total update.
  lr = cfg.optim.lr
  for (k, p), (rk, rp) in zip(model.named_parameters(), ref.named_parameters()):
    if not p.requires_grad:
      continue
    d = (p.detach().cpu() - rp.detach()).abs().max().item()
    assert d <= 0.05 * lr * steps + 1e-7, f'{k}: parameter drift {d:.3e} after {steps} steps'
  for s, rs in zip(state['ema'].shadow_params, rstate['ema'].shadow_params):
    assert (s.detach().cpu() - rs).abs().max().item() <= 0.05 * lr * steps + 1e-7
  return state, rstate


def loss_curve(st, lib, family='vp', steps=100, B=8, tol=1e-3):
  """north_star "loss curve matching reference within tolerance": `steps` consecutive step_fn calls (losses.py:262-293:
  t_min draw, importance-sampled times, loss, backward, warm-up, clip, Adam, EMA) on the product engine and on RefNet +
  torch.optim.Adam from the same weights, batches and noise.  Three steps prove the update rule; a trajectory shows that the
  ~1e-6 per-step differences do not compound through Adam's 1 / (sqrt(v) + 1e-8): per-sample losses within `tol` (1e-3
  relative, the north star's fp32 bar) at EVERY step, the batch-mean curve within tol, parameters within 5 % of the total
  update at the end."""
  base = tiny_config(st, family)
  base.optim.warmup = 10
  cfg, cfg_cpu, sde, model, ref = build_pair(st, base, lib)
  dev = cfg.device
  state, rstate = make_state(st, cfg, model), make_state(st, cfg_cpu, ref)
  state['optimizer']._backend = lib
  state['ema'].set_backend(lib)
  step_fn = st.losses.get_step_fn(cfg, sde, train=True, optimize_fn=st.losses.optimization_manager(cfg))
  rstep_fn = st.losses.get_step_fn(cfg_cpu, sde, train=True, optimize_fn=st.losses.optimization_manager(cfg_cpu))
  curve, rcurve, worst = [], [], 0.0
  threads = torch.get_num_threads()
  torch.set_num_threads(min(threads, 8))      # the fixture-size RefNet on a 128-thread host spends its time in thread hand-offs
  for i in range(steps):
    batch = st.datasets.synthetic_batch(cfg_cpu, B, generator=torch.Generator().manual_seed(1000 + i))
    np.random.seed(70 + i)
    with patched_rng(500 + i):
      loss = step_fn(state, batch.to(dev))
    np.random.seed(70 + i)
    with patched_rng(500 + i):
      rloss = rstep_fn(rstate, batch)
    e = rel_err(loss, rloss)
    worst = max(worst, e)
    assert e <= tol, f'step {i}: per-sample losses differ by {e:.3e}'
    curve.append(loss.mean().item()); rcurve.append(rloss.mean().item())
  torch.set_num_threads(threads)
  curve, rcurve = np.array(curve), np.array(rcurve)
  mean_err = np.abs(curve - rcurve).max() / np.abs(rcurve).max()
  assert mean_err <= tol, f'loss curve differs by {mean_err:.3e}'
  lr = cfg.optim.lr
  drift = 0.0
  for (k, p), (rk, rp) in zip(model.named_parameters(), ref.named_parameters()):
    if p.requires_grad:
      drift = max(drift, (p.detach().cpu() - rp.detach()).abs().max().item())
  assert drift <= 0.05 * lr * steps, f'parameter drift {drift:.3e} after {steps} steps'
  return dict(steps=steps, worst_per_sample=worst, curve_err=float(mean_err), first=float(rcurve[0]), last=float(rcurve[-1]), drift=drift)


def dropout_consistency(st, lib):
  """With dropout on, forward and backward must use the SAME mask: check the gradient of a training
  forward by finite differences through the frozen mask (same seed via the same torch CPU draw)."""
  cfg, cfg_cpu, sde, model, ref = build_pair(st, tiny_config(st, 'vp', dropout=0.3), lib)
  dev = cfg.device
  x, t, cond = _inputs(cfg, sde, 2)
  model.train()
  go = torch.randn(x.shape, generator=torch.Generator().manual_seed(5)).to(dev)

  def f(xin):
    torch.manual_seed(123)            # the engine draws its per-call dropout seed from torch's CPU generator
    return (model(xin, cond.to(dev)) * go).sum()

  xg = x.clone().to(dev).requires_grad_(True)
  f(xg).backward()
  v = torch.randn(x.shape, generator=torch.Generator().manual_seed(6)).to(dev)
  eps = 1e-2
  with torch.no_grad():
    fd = (f(xg.detach() + eps * v) - f(xg.detach() - eps * v)).item() / (2 * eps)
  an = (xg.grad * v).sum().item()
  assert abs(fd - an) <= 2e-2 * max(abs(an), abs(fd), 1e-3), (fd, an)
  # and eval mode ignores dropout entirely
  model.eval(); ref.eval()
  with torch.no_grad():
    assert rel_err(model(x.to(dev), cond.to(dev)), ref(x, cond)) <= TOL


def pc_sampler_steps(st, lib, family, n=3):
  """A few iterations of the PC loop + the denoising step, product vs oracle, same noise."""
  base = tiny_config(st, family)
  if family == 'vp':
    base.sampling.method, base.sampling.predictor, base.sampling.corrector = 'pc', 'euler_maruyama', 'none'
  elif family == 've':
    base.sampling.method, base.sampling.predictor, base.sampling.corrector = 'pc', 'reverse_diffusion', 'langevin'
  else:
    raise ValueError('RVE PC sampling raises in the reference (SURVEY.md a6)')
  cfg, cfg_cpu, sde, model, ref = build_pair(st, base, lib)
  sde.N = n      # shorten the loop (the sigma / beta ladders keep their configured length)
  shape = (2, cfg.data.num_channels, cfg.data.image_size, cfg.data.image_size)
  inv = st.datasets.get_data_inverse_scaler(cfg)
  fn = st.sampling.get_sampling_fn(cfg, sde, shape, inv, 1e-3)
  rfn = st.sampling.get_sampling_fn(cfg_cpu, sde, shape, inv, 1e-3)
  with patched_rng(11):
    xs, nfe = fn(model)
  with patched_rng(11):
    xr, rnfe = rfn(ref)
  assert nfe == rnfe
  assert rel_err(xs, xr) <= 5 * TOL, f'sampler mismatch {rel_err(xs, xr):.3e}'


def pc_sampler_full_length(st, lib, family, N=None, B=2, tol=1e-3, seed=11):
  """The WHOLE predictor-corrector trajectory (sampling.py:365-433: the config's own N -- 1000 for VP, 2000 for the VE
  nets -- from t = T down to eps, corrector first, then predictor, then the denoising step), product engine against RefNet in
  lockstep: both sides run the reference's update rules on their own state with the SAME injected noise, and the relative
  difference of the two states is recorded after every iteration (sigma runs from 348 down to 0.01 on the VE ladder, so the
  network is evaluated over its whole conditioning range).  Final samples within `tol`; the lockstep product trajectory
  must also be what get_sampling_fn()'s own loop returns, bit for bit.  N: shorten the grid (CPU suite only)."""
  from _model_util import _Draws
  base = tiny_config(st, family)
  if family == 'vp':
    base.sampling.method, base.sampling.predictor, base.sampling.corrector = 'pc', 'euler_maruyama', 'none'
  else:
    base.sampling.method, base.sampling.predictor, base.sampling.corrector = 'pc', 'reverse_diffusion', 'langevin'
  cfg, cfg_cpu, sde, model, ref = build_pair(st, base, lib)
  if N is not None:
    sde.N = N
  S = st.sampling
  shape = (B, cfg.data.num_channels, cfg.data.image_size, cfg.data.image_size)
  inv = st.datasets.get_data_inverse_scaler(cfg)
  s = cfg.sampling
  pred, corr = S.get_predictor(s.predictor), S.get_corrector(s.corrector)
  eps = 1e-3
  names = ('rand', 'randn', 'randn_like', 'randint_like')

  class using:
    def __init__(self, d):
      self.d = d
    def __enter__(self):
      self.saved = [getattr(torch, n) for n in names]
      for n in names:
        setattr(torch, n, getattr(self.d, n))
    def __exit__(self, *a):
      for n, f in zip(names, self.saved):
        setattr(torch, n, f)

  sides = []
  for c, m in ((cfg, model), (cfg_cpu, ref)):
    d = _Draws(seed)
    with using(d):
      x = sde.prior_sampling(shape).to(c.device)
    sides.append(dict(cfg=c, model=m, draws=d, x=x, x_mean=x, grid=torch.linspace(sde.T, eps, sde.N, device=c.device)))
  errs = []
  threads = torch.get_num_threads()
  torch.set_num_threads(min(threads, 8))
  try:
    with torch.no_grad():
      for i in range(sde.N):
        for sd in sides:
          c = sd['cfg']
          vec_t = torch.ones(shape[0], device=c.device) * sd['grid'][i]
          with using(sd['draws']):
            x, x_mean = S.shared_corrector_update_fn(sd['x'], vec_t, sde, sd['model'], corr, c.training.continuous, s.snr,
                                                     s.n_steps_each, c)
            x, x_mean = S.shared_predictor_update_fn(x, vec_t, sde, sd['model'], pred, s.probability_flow,
                                                     c.training.continuous, c)
          sd['x'], sd['x_mean'] = x, x_mean
        errs.append(rel_err(sides[0]['x'], sides[1]['x']))
        assert np.isfinite(errs[-1]) and errs[-1] <= 10 * tol, f'iteration {i}: states differ by {errs[-1]:.3e}'
      outs = []
      for sd in sides:
        with using(sd['draws']):
          xm = S._denoiser(sd['cfg'], sde, probability_flow=True)(sd['model'], sd['x_mean'] if s.noise_removal else sd['x'])
        outs.append(inv(xm))
  finally:
    torch.set_num_threads(threads)
  final = rel_err(outs[0], outs[1])
  assert final <= tol, f'final samples differ by {final:.3e} after {sde.N} iterations'
  # the sampler's own loop (frozen weights, tqdm) gives the lockstep trajectory's result
  with patched_rng(seed):
    xs, nfe = S.get_sampling_fn(cfg, sde, shape, inv, eps)(model)
  assert nfe == sde.N * (s.n_steps_each + 1)
  assert torch.equal(xs, outs[0]), float((xs - outs[0]).abs().max())
  errs = np.array(errs)
  k = max(1, sde.N // 10)
  return dict(N=sde.N, nfe=nfe, final=final, worst_iteration=float(errs.max()), at=int(errs.argmax()),
              trace=[float(f'{e:.2e}') for e in errs[::k]])


def ode_sampler(st, lib, tol=1e-3):
  base = tiny_config(st, 'vp')
  cfg, cfg_cpu, sde, model, ref = build_pair(st, base, lib)
  shape = (2, 3, cfg.data.image_size, cfg.data.image_size)
  inv = st.datasets.get_data_inverse_scaler(cfg)
  fn = st.sampling.get_ode_sampler(cfg, sde, shape, inv, denoise=True, rtol=tol, atol=tol, eps=1e-3, device=cfg.device)
  rfn = st.sampling.get_ode_sampler(cfg_cpu, sde, shape, inv, denoise=True, rtol=tol, atol=tol, eps=1e-3, device='cpu')
  with patched_rng(3):
    xs, nfe = fn(model)
  with patched_rng(3):
    xr, rnfe = rfn(ref)
  # Both sides run the same device RK45 (engine/rk45.py, pinned to SciPy's) on networks that agree to ~1e-6: the
  # controller takes the same decisions, so the step sequences are identical and the states agree far below the
  # solver tolerance.
  assert nfe == rnfe, (nfe, rnfe)
  assert rel_err(xs, xr) <= 1e-3, rel_err(xs, xr)
  return dict(rtol=tol, nfe=nfe, err=rel_err(xs, xr))


def checkpoint_roundtrip(st, lib, tmp_path):
  state, rstate = train_steps(st, lib, 'vp', steps=2)
  cfg = state['model'].module.config
  path = str(tmp_path / 'checkpoint.pth')
  st.utils.save_checkpoint(cfg, path, state)
  saved = torch.load(path, weights_only=False)
  assert set(saved) == {'optimizer', 'model', 'ema', 'step'}
  assert all(k.startswith('module.') for k in saved['model'])
  assert set(saved['ema']) == {'decay', 'num_updates', 'shadow_params'}
  # the optimizer state has torch.optim.Adam's structure, so it loads into torch's Adam ...
  rstate['optimizer'].load_state_dict(saved['optimizer'])
  # ... and a fresh product state restores everything
  cfg2, _, sde, model2, _ = build_pair(st, tiny_config(st, 'vp'), lib, seed=9)
  state2 = make_state(st, cfg2, model2)
  state2['optimizer']._backend = lib
  state2['ema'].set_backend(lib)
  st.utils.restore_checkpoint(cfg2, path, state2, cfg2.device)
  assert state2['step'] == 2 and state2['optimizer']._step == 2
  for p, q in zip(state['model'].parameters(), state2['model'].parameters()):
    assert torch.equal(p.detach().cpu(), q.detach().cpu())
  assert torch.equal(state['optimizer']._m.cpu(), state2['optimizer']._m.cpu())
  for a, b in zip(state['ema'].shadow_params, state2['ema'].shadow_params):
    assert torch.equal(a.cpu(), b.cpu())


def golden_forward_backward(st, lib, family):
  """Product path against the committed fixtures generated from the reference itself."""
  import os
  g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', f'model_{family}.npz')))
  base = {'vp': st.configs.cifar10_ddpmpp_nll_st, 'rve': st.configs.celeba_uncsnpp_st,
          've': st.configs.celebahq_uncsnpp_st}[family]()
  cfg = st.configs.tiny(base, nf=8, ch_mult=(1, 1, 2) if family == 've' else (1, 2), num_res_blocks=1,
                        image_size=8, attn_resolutions=(4,), dropout=0.0)
  dev = torch.device('cuda:0') if lib.is_device else torch.device('cpu')
  cfg.device = dev
  net = st.models.ncsnpp.NCSNpp(cfg, None)
  net.load_state_dict({k[3 + len('module.'):]: torch.from_numpy(v) for k, v in g.items() if k.startswith('sd.')})
  net.set_backend(lib)
  net = net.to(dev).eval()
  x, cond = torch.from_numpy(g['x']).to(dev), torch.from_numpy(g['cond']).to(dev)
  xr = x.clone().requires_grad_(True)
  y = net(xr, cond)
  assert rel_err(y, torch.from_numpy(g['net'])) <= TOL
  (y * torch.from_numpy(g['go']).to(dev)).sum().backward()
  assert rel_err(xr.grad, torch.from_numpy(g['gx'])) <= TOL
  grads = dict(net.named_parameters())
  scale = float(np.sqrt(g['grad_sumsq']))
  for n in g['grad_names']:
    got = grads[str(n)[len('module.'):]].grad.detach().cpu().numpy()
    assert np.abs(got - g['grad.' + str(n)]).max() <= TOL * scale, n
  sde = st.sde_lib.get_sde(cfg, None)
  model = st.models.utils.DataParallel(net)
  s = st.models.utils.get_score_fn(cfg, sde, model, train=False, continuous=True)(x, torch.from_numpy(g['t']).to(dev))
  assert rel_err(s, torch.from_numpy(g['score'])) <= TOL


def golden_likelihood(st, model, cfg, family, tol, nfe_exact=True):
  """likelihood.py (residual term, ST-NELBO, ODE NLL) on `model` against the reference's own outputs
  (tests/golden/likelihood_{family}.npz, generated by tools/make_golden.py; noise injected)."""
  import os
  from _model_util import patched_rng
  g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', f'likelihood_{family}.npz')))
  dev = next(model.parameters()).device
  sde = st.sde_lib.get_sde(cfg, None)
  cfg.eval.probability_flow, cfg.eval.lambda_ = True, 0.0
  data = torch.from_numpy(g['data']).to(dev)
  inv = st.datasets.get_data_inverse_scaler(cfg)
  model.eval()
  score_fn = st.models.utils.get_score_fn(cfg, sde, model, train=False, continuous=True)

  def close(a, ref, what):
    a = a.detach().cpu().double().numpy()
    assert np.abs(a - ref).max() <= tol * max(np.abs(ref).max(), 1.0), (what, a, ref)

  for deq in ('none', 'lossless'):
    cfg.data.dequantization = deq
    for var in ('ddpm', 'scoreflow'):
      with patched_rng(21), torch.no_grad():
        r = st.likelihood.get_likelihood_residual_fn(cfg, sde, score_fn, variance=var)(data, 1e-3)
      close(r, g[f'residual.{deq}.{var}'], f'residual {deq} {var}')
  cfg.data.dequantization = 'none'
  np.random.seed(3)
  with patched_rng(31):
    nelbo, resid = st.likelihood.get_elbo_fn(cfg, sde, inverse_scaler=inv)(model, data, eps=1e-3)
  close(nelbo, g['nelbo'], 'nelbo')
  close(resid, g['nelbo.residual'], 'nelbo residual')
  with patched_rng(41):
    bpd, z, nfe = st.likelihood.get_likelihood_fn(cfg, sde, inv)(model, data, eps=1e-3)
  if nfe == int(g['nll.nfe']):
    close(bpd, g['nll.bpd'], 'nll bpd')
    close(z, g['nll.z'], 'nll latent')
  else:
    # An adaptive solver amplifies rounding-level differences of the network into a different step sequence when an
    # error estimate lands next to the acceptance threshold; the two solutions then agree to the solver tolerance
    # (rtol = atol = 1e-5, the reference's defaults), not to the arithmetic tolerance.
    assert not nfe_exact, (nfe, int(g['nll.nfe']))
    assert abs(nfe - int(g['nll.nfe'])) <= 0.1 * int(g['nll.nfe']), (nfe, int(g['nll.nfe']))
    b, zz = bpd.detach().cpu().double().numpy(), z.detach().cpu().double().numpy()
    assert np.abs(b - g['nll.bpd']).max() <= 1e-3 * np.abs(g['nll.bpd']).max(), ('nll bpd', b, g['nll.bpd'])
    assert np.abs(zz - g['nll.z']).max() <= 1e-2 * np.abs(g['nll.z']).max(), 'nll latent'


def golden_likelihood_product(st, lib, family):
  """The HIP engine (or the checker) through likelihood.py against the reference's likelihood fixtures: needs the
  network's input gradient (Hutchinson divergence, NELBO Jacobian-vector product) from the hand-written backward."""
  import os
  g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', f'model_{family}.npz')))
  base = {'vp': st.configs.cifar10_ddpmpp_nll_st, 've': st.configs.celebahq_uncsnpp_st}[family]()
  cfg = st.configs.tiny(base, nf=8, ch_mult=(1, 1, 2) if family == 've' else (1, 2), num_res_blocks=1,
                        image_size=8, attn_resolutions=(4,), dropout=0.0)
  dev = torch.device('cuda:0') if lib.is_device else torch.device('cpu')
  cfg.device = dev
  net = st.models.ncsnpp.NCSNpp(cfg, None)
  net.load_state_dict({k[3 + len('module.'):]: torch.from_numpy(v) for k, v in g.items() if k.startswith('sd.')})
  net.set_backend(lib)
  model = st.models.utils.DataParallel(net.to(dev).eval())
  # fp32 kernels vs the reference's CPU arithmetic: the adaptive solver may take a different number of steps only if
  # an error estimate lands within rounding of the acceptance threshold; the fixture's tolerance keeps it identical
  golden_likelihood(st, model, cfg, family, tol=TOL, nfe_exact=False)


def prepared_weights_coherence(st, lib):
  """Engine-level contract of the prepared convolution weights (engine/executor.py): a forward always sees the current
  parameters -- after an in-place update (optimizer step, EMA swap) the batched preparation runs again -- and the
  results are bit-identical to per-call preparation (STK_WP=0), forward and backward.  Uses a net wide enough
  (96 channels) for the split kernels; the tiny parity configs never reach them."""
  base = st.configs.cifar10_ddpmpp_nll_st()
  cfg = st.configs.tiny(base, nf=96, ch_mult=(1,), num_res_blocks=1, image_size=16, attn_resolutions=(), dropout=0.0)
  dev = torch.device('cuda:0')
  cfg.device = dev
  torch.manual_seed(3)
  net = st.models.ncsnpp.NCSNpp(cfg, None)
  net.set_backend(lib)
  net = net.to(dev).eval()
  ex = net.engine()
  g = torch.Generator().manual_seed(7)
  x = torch.randn(12, 3, 16, 16, generator=g).to(dev)
  cond = (torch.rand(12, generator=g) * 999).to(dev)

  def fwd():
    with torch.no_grad():
      return net(x, cond)

  def fwd_bwd():
    xi = x.clone().requires_grad_(True)
    net.zero_grad()
    (net(xi, cond) ** 2).sum().backward()
    return xi.grad.clone(), torch.cat([p.grad.reshape(-1) for p in net.parameters() if p.grad is not None])

  assert ex.use_wp
  y1 = fwd()
  progs = list(ex.programs.values())
  assert progs and progs[0].wp_counts[0] > 0, 'no layer of this net took the prepared-weight path'
  for p in net.parameters():
    p.data.mul_(1.25)
  y2 = fwd()
  gx2, gw2 = fwd_bwd()
  assert progs[0].wp_counts[1] > progs[0].wp_counts[0] or any(pr.wp_counts[1] > pr.wp_counts[0] for pr in ex.programs.values())
  ex.use_wp = False                    # per-call preparation of the same (updated) weights
  y3 = fwd()
  gx3, gw3 = fwd_bwd()
  ex.use_wp = True
  assert not torch.equal(y1, y2)
  assert torch.equal(y2, y3), 'a forward after an in-place parameter update used stale prepared weights'
  assert torch.equal(gx2, gx3), float((gx2 - gx3).abs().max())
  assert torch.equal(gw2, gw3), (float((gw2 - gw3).abs().max()), float(gw2.abs().max()), float((gw3 / gw2).nanmedian()))
  with ex.frozen_weights():            # promise kept: same values, one preparation
    ya, yb = fwd(), fwd()
  assert torch.equal(ya, y2) and torch.equal(yb, y2)
  for p in net.parameters():
    p.data.mul_(0.8)
  y4 = fwd()                           # outside the block the next forward prepares again
  ex.use_wp = False
  y5 = fwd()
  ex.use_wp = True
  assert torch.equal(y4, y5) and not torch.equal(y4, y2)


SAMPLER_CASES = [('vp', 'reverse_diffusion', 'langevin'), ('vp', 'ancestral_sampling', 'ald'),
                 ('vp', 'euler_maruyama', 'langevin'), ('ve', 'ancestral_sampling', 'ald'),
                 ('ve', 'euler_maruyama', 'none')]


def golden_sampler_registry(st, make_model, tol, bit_exact_sde=False):
  """Every predictor / corrector of the registry beyond the configs' default pairs, plus the sub-VP SDE, against the
  reference's own outputs (tests/golden/samplers.npz, tools/make_golden.py: samplers_fixture).
  make_model(family) -> (cfg, model) with the weights of model_{family}.npz on the backend under test."""
  import os
  g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'samplers.npz')))
  worst = {}
  for family, pred, corr in SAMPLER_CASES:
    cfg, model = make_model(family)
    cfg.sampling.method, cfg.sampling.predictor, cfg.sampling.corrector = 'pc', pred, corr
    sde = st.sde_lib.get_sde(cfg, None)
    sde.N = 3
    H = cfg.data.image_size
    fn = st.sampling.get_sampling_fn(cfg, sde, (2, 3, H, H), st.datasets.get_data_inverse_scaler(cfg), 1e-3)
    with patched_rng(13):
      xs, nfe = fn(model)
    key = f'{family}.default.{pred}.{corr}'
    assert nfe == int(g[key + '.nfe']), key
    ref = g[key + '.samples']
    err = np.abs(xs.detach().cpu().numpy() - ref).max() / np.abs(ref).max()
    worst[key] = err
    assert err <= tol, f'{key}: {err:.3e}'
  # sub-VP: the SDE's own functions bit for bit, and single predictor updates through the network
  cfg, model = make_model('vp')
  cfg.training.sde = 'subvpsde'
  sde = st.sde_lib.get_sde(cfg, None)
  x, t = torch.from_numpy(g['subvp.x']), torch.from_numpy(g['subvp.t'])
  mean, std = sde.marginal_prob(x, t)
  drift, diff = sde.sde(x, t)
  f, G = sde.discretize(x, t)
  for k, v in (('mean', mean), ('std', std), ('drift', drift), ('diffusion', diff), ('prior_logp', sde.prior_logp(x)),
               ('disc_f', f), ('disc_G', G)):
    # host arithmetic: bit-identical on the machine that generated the fixture (CPU suite); another host's libm /
    # vector width may differ in the last place (the GPU box's CPU)
    if bit_exact_sde:
      assert np.array_equal(v.numpy(), g['subvp.' + k]), 'subvp.' + k
    else:
      assert np.allclose(v.numpy(), g['subvp.' + k], rtol=2e-6, atol=0), 'subvp.' + k
  dev = next(model.parameters()).device
  xs, ts = torch.from_numpy(g['subvp.pred.x']).to(dev), torch.from_numpy(g['subvp.pred.t']).to(dev)
  model.eval()
  with torch.no_grad():
    s = st.models.utils.get_score_fn(cfg, sde, model, train=False, continuous=True)(xs, ts)
  assert rel_err(s, torch.from_numpy(g['subvp.score'])) <= tol
  for pred in ('euler_maruyama', 'reverse_diffusion'):
    with patched_rng(17), torch.no_grad():
      xn, xm = st.sampling.shared_predictor_update_fn(xs, ts, sde, model, st.sampling.get_predictor(pred), False, True, cfg)
    assert rel_err(xn, torch.from_numpy(g[f'subvp.pred.{pred}.x'])) <= tol, pred
    assert rel_err(xm, torch.from_numpy(g[f'subvp.pred.{pred}.x_mean'])) <= tol, pred
  return worst


def golden_sampler_registry_product(st, lib):
  import os
  def make(family):
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', f'model_{family}.npz')))
    base = {'vp': st.configs.cifar10_ddpmpp_nll_st, 've': st.configs.celebahq_uncsnpp_st}[family]()
    cfg = st.configs.tiny(base, nf=8, ch_mult=(1, 1, 2) if family == 've' else (1, 2), num_res_blocks=1,
                          image_size=8, attn_resolutions=(4,), dropout=0.0)
    dev = torch.device('cuda:0') if lib.is_device else torch.device('cpu')
    cfg.device = dev
    net = st.models.ncsnpp.NCSNpp(cfg, None)
    net.load_state_dict({k[3 + len('module.'):]: torch.from_numpy(v) for k, v in g.items() if k.startswith('sd.')})
    net.set_backend(lib)
    return cfg, st.models.utils.DataParallel(net.to(dev).eval())
  return golden_sampler_registry(st, make, TOL)


def _stream_beside(dev):
  """A pooled stream that really runs beside the current one: every fourth-to-eighth pooled stream shares the current
  stream's hardware queue and would run strictly after it (engine/executor.py: checked_side_stream) -- a neighbour on such a
  stream is no neighbour."""
  from importlib import import_module
  ex = import_module('soft-truncation_amd.engine.executor')
  main = torch.cuda.current_stream(dev)
  s = None
  for _ in range(16):
    s = torch.cuda.Stream(dev)
    if ex._overlap_ratio(main, s) < 1.5:
      break
  return s


class _MfmaNeighbour:
  """A kernel of ANOTHER stream that issues MFMAs into accumulation registers beside whatever runs meanwhile: the library's
  own fp32-operand weight gradient of a 192 -> 192 3x3 layer at 8x8 (x2::wgrad3_kernel), launched `calls` times on a
  stream of its own.  This is the neighbour that made a packed-fp32 instruction of the GroupNorm backward return a wrong
  16-lane slice in round 3 (DESIGN.md "The hazard"; tools/_probe/cores.hip reproduces it with exactly this call)."""

  def __init__(self, lib, dev, N=96, C=192, H=8):
    g = torch.Generator().manual_seed(11)
    self.lib, self.N, self.C, self.H = lib, N, C, H
    self.x = torch.randn(N, C, H, H, generator=g).to(dev)
    self.dy = (0.01 * torch.randn(N, C, H, H, generator=g)).to(dev)
    self.dw = torch.zeros(C, C, 3, 3, device=dev)
    self.ws_bytes = int(lib.conv2d_wgrad_ws_bytes(C, 0, N, C, H, H, 3, 3))
    self.ws = torch.empty(self.ws_bytes // 4 + 64, device=dev)
    self.stream = _stream_beside(dev)

  def launch(self, calls=40):
    s = self.stream
    for _ in range(calls):
      self.lib.conv2d_wgrad_f32(self.x.data_ptr(), self.C, None, 0, self.dy.data_ptr(), self.dw.data_ptr(), 0, 1.0,
                                self.ws.data_ptr(), self.ws_bytes, self.N, self.H, self.H, self.C, self.H, self.H, 3, 3, 1, 1,
                                s.cuda_stream)


class _CopyNeighbour:
  """What a gradient exchange puts beside the backward on a multi-GPU run, as far as one GPU can show it: a stream of its own
  that moves 64 MB blocks (the bucket size of engine/ddp.py) through HBM -- reduce-copy kernels of a ring step read two
  buffers and write one -- while the backward runs."""

  def __init__(self, dev, mb=64):
    n = mb * 2 ** 20 // 4
    self.a = torch.randn(n, device=dev)
    self.b = torch.randn(n, device=dev)
    self.c = torch.empty(n, device=dev)
    self.stream = _stream_beside(dev)

  def launch(self, calls=24):
    with torch.cuda.stream(self.stream):
      for _ in range(calls):
        torch.add(self.a, self.b, out=self.c)


def two_streams_deterministic(st, lib, B=96, delays=(150000, 600000, 2000000), neighbour_runs=150, copy_runs=50):
  """The backward with its weight gradients / shortcut convolutions on the side stream (engine/executor.SideStream) must
  give the gradients of a quiet one-stream backward BIT FOR BIT, whatever else is resident on the chip:
    * every side launch site in turn held back by a spin kernel (torch.cuda._sleep) of three lengths, with and without
    * a neighbour stream that issues MFMAs into accumulation registers throughout the backward (_MfmaNeighbour), which
      also runs beside the plain two-stream and the ONE-stream backward `neighbour_runs` times each;
    * a neighbour stream that streams 64 MB blocks through HBM like the reduce-copy kernels of a gradient exchange
      (_CopyNeighbour), `copy_runs` times beside each of the two backward modes.
  Watch on the packed-fp32 hazard (csrc/Makefile HAZARD_FLAGS, profiles/r04_pk_hazard.txt): a build with the SLP vectoriser
  on fails every one of these runs (tools/_probe/side_race2.py: 168 of 168)."""
  from importlib import import_module
  G = import_module('soft-truncation_amd.engine.graph')
  cfg, cfg_cpu, sde, model, ref = build_pair(st, tiny_config(st, 'wide'), lib)
  ex = model.module.engine()
  if not ex.use_side:
    pytest.skip('side stream switched off (STK_WGRAD_STREAM=0)')
  dev = cfg.device
  x, t, cond = _inputs(cfg, sde, B)
  go = torch.randn(B, *x.shape[1:], generator=torch.Generator().manual_seed(5)).to(dev)
  model.eval()
  nb = _MfmaNeighbour(lib, dev)
  cp = _CopyNeighbour(dev)

  def run(side, neighbour=False):
    if neighbour == 'copy':
      ex.use_side = side
      model.zero_grad()
      xg = x.clone().to(dev).requires_grad_(True)
      y = model(xg, cond.to(dev))
      cp.launch()
      (y * go).sum().backward()
      busy = not cp.stream.query()
      torch.cuda.synchronize()
      return [p.grad.detach().clone() for p in model.parameters()] + [xg.grad.clone()], busy
    ex.use_side = side
    model.zero_grad()
    xg = x.clone().to(dev).requires_grad_(True)
    y = model(xg, cond.to(dev))
    if neighbour:
      nb.launch()
    (y * go).sum().backward()
    busy = neighbour and not nb.stream.query()          # still running when the backward had been queued
    torch.cuda.synchronize()
    return [p.grad.detach().clone() for p in model.parameters()] + [xg.grad.clone()], busy

  saved = (G._SIDE_DELAY, G._SIDE_DELAY_FILTER)
  try:
    base, _ = run(False)
    sites = []
    G._SIDE_DELAY, G._SIDE_DELAY_FILTER = 1, (lambda n: (sites.append(n), False)[1])
    run(True)
    sites = list(dict.fromkeys(sites))
    assert len(sites) >= 10, sites
    runs = bad = beside = 0

    def check(got):
      nonlocal runs, bad, beside
      runs += 1
      beside += bool(got[1])
      bad += any(not torch.equal(a, b) for a, b in zip(got[0], base))

    for neighbour in (False, True):
      for delay in delays:
        for site in sites:
          G._SIDE_DELAY, G._SIDE_DELAY_FILTER = delay, (lambda n, site=site: n == site)
          check(run(True, neighbour))
    G._SIDE_DELAY, G._SIDE_DELAY_FILTER = 0, None
    for side in (True, False):
      for _ in range(neighbour_runs):
        check(run(side, True))
      for _ in range(copy_runs):
        check(run(side, 'copy'))
    assert bad == 0, f'{bad} of {runs} backward passes differ from the quiet one-stream result'
    assert beside >= neighbour_runs, f'the neighbour kernels ended too early to be resident beside the backward ({beside} of {runs})'
  finally:
    G._SIDE_DELAY, G._SIDE_DELAY_FILTER = saved
    ex.use_side = True
  return runs


def score_matching_pieces(st, lib):
  """losses._ScoreMatching on samples larger than its PIECE (3 x 128 x 128 = 49152 elements: cut into pieces that the kernel
  sees as samples of their own and that are summed afterwards, so the summation order differs from torch.mean / torch.sum):
  forward and backward against the torch expressions of losses.py:122-132, reduce_mean on and off, VP (score = -net / std)
  and VE (score = net) forms."""
  dev = torch.device('cuda:0') if lib.is_device else torch.device('cpu')
  g = torch.Generator().manual_seed(9)
  B, shape = 5, (3, 128, 128)
  inner = 3 * 128 * 128
  SM = st.losses._ScoreMatching
  assert SM._pieces(B, inner) > 1, 'the case must exercise the piece-split branch'
  out = {}
  for vp in (True, False):
    for reduce_mean in (True, False):
      net = torch.randn(B, *shape, generator=g).to(dev).requires_grad_(True)
      z = torch.randn(B, *shape, generator=g).to(dev)
      std = (torch.rand(B, generator=g) * 3 + 0.05).to(dev)
      wgt = (torch.rand(B, generator=g) + 0.5).to(dev)
      go = torch.randn(B, generator=g).to(dev)
      losses = SM.apply(net, z, std, wgt, lib, vp, reduce_mean)
      (losses * go).sum().backward()
      netr = net.detach().clone().requires_grad_(True)
      s4 = std[:, None, None, None]
      score = -netr / s4 if vp else netr
      r2 = torch.square(score * s4 + z).reshape(B, -1)
      want = wgt * (torch.mean(r2, dim=-1) if reduce_mean else 0.5 * torch.sum(r2, dim=-1))
      (want * go).sum().backward()
      e_l = float(((losses.detach() - want.detach()).abs() / want.detach().abs()).max())
      e_g = float((net.grad - netr.grad).abs().max() / netr.grad.abs().max())
      out[(vp, reduce_mean)] = (e_l, e_g)
      assert e_l <= 2e-6 and e_g <= 1e-5, (vp, reduce_mean, e_l, e_g)
  return out


def fused_loss_matches_torch(st, lib, family):
  """losses.get_sde_loss_fn: the three-kernel path (stk_perturb / network / stk_sm_loss) against the reference's torch
  expressions on the SAME engine, noise and times: x_t bit-identical, per-sample losses to 1e-6, gradients to 1e-5."""
  cfg, cfg_cpu, sde, model, ref = build_pair(st, tiny_config(st, family), lib)
  dev = cfg.device
  B = 6
  batch = st.datasets.synthetic_batch(cfg_cpu, B, generator=torch.Generator().manual_seed(3)).to(dev)
  loss_fn = st.losses.get_sde_loss_fn(cfg, sde, train=True)
  out = {}
  saved = st.losses.FUSED_LOSS
  try:
    for fused in (True, False):
      st.losses.FUSED_LOSS = fused
      model.zero_grad()
      np.random.seed(0)
      with patched_rng(17):
        losses = loss_fn(model, batch, importance_sampling=cfg.training.importance_sampling, t_min=1e-3)
      torch.mean(losses).backward()
      out[fused] = (losses.detach().cpu().double(), [p.grad.detach().cpu().double().clone() for p in model.parameters() if p.grad is not None])
  finally:
    st.losses.FUSED_LOSS = saved
  (lf, gf), (lt, gt) = out[True], out[False]
  e_loss = float(((lf - lt).abs() / lt.abs()).max())
  scale = max(float(g.abs().max()) for g in gt)
  e_grad = max(float((a - b).abs().max()) for a, b in zip(gf, gt)) / scale
  assert e_loss <= 1e-6 and e_grad <= 1e-5, (e_loss, e_grad)
  # the perturbation kernel against the torch expression, bit for bit (device tensors on the HIP library)
  g = torch.Generator().manual_seed(5)
  x = torch.randn(B, 3, 16, 16, generator=g).to(dev)
  z = torch.randn(B, 3, 16, 16, generator=g).to(dev)
  a = torch.rand(B, generator=g).to(dev)
  s_ = (torch.rand(B, generator=g) * 3).to(dev)
  from _util import call
  got = torch.empty_like(x)
  call(lib, 'perturb_f32', x, z, a, s_, got, B, x[0].numel())
  want = a[:, None, None, None] * x + s_[:, None, None, None] * z
  got_ve = torch.empty_like(x)
  call(lib, 'perturb_f32', x, z, None, s_, got_ve, B, x[0].numel())
  want_ve = x + s_[:, None, None, None] * z
  if lib.is_device:
    assert torch.equal(got, want) and torch.equal(got_ve, want_ve)
  else:
    assert (got - want).abs().max().item() <= 1e-6 and (got_ve - want_ve).abs().max().item() <= 1e-6
  return e_loss, e_grad
