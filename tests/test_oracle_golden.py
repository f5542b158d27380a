"""Pin the oracle (and the host-side restatements that travel with it) to the REFERENCE.

tests/golden/*.npz were produced by tools/make_golden.py, which runs the reference's own PyTorch CPU
path (imported from /root/reference) on seeded inputs with every noise draw made explicit.  These tests
run anywhere (no reference, no GPU):

  * sde_lib                      == reference sde_lib            (bit-identical)
  * oracle/ref_torch.RefNet      == reference NCSNpp             (forward, score, gradients)
  * losses.get_step_fn / sampling on RefNet + torch Adam == the reference's step_fn / PC sampler
  * oracle/stk_ref.c (the C checker of the HIP kernels) == the reference's upfirdn2d_native etc.
  * the planned-graph engine on the C checker == the reference network (host logic)
"""
import copy
import os

import numpy as np
import pytest
import torch

import ref_torch
from _model_util import patched_rng
from _util import call

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
  return dict(np.load(os.path.join(GOLD, name), allow_pickle=False))


def tiny_cfg(st, family):
  base = {'vp': st.configs.cifar10_ddpmpp_nll_st, 'rve': st.configs.celeba_uncsnpp_st,
          've': st.configs.celebahq_uncsnpp_st}[family]()
  cfg = st.configs.tiny(base, nf=8, ch_mult=(1, 1, 2) if family == 've' else (1, 2), num_res_blocks=1,
                        image_size=8, attn_resolutions=(4,), dropout=0.0)
  cfg.optim.warmup = 2
  cfg.device = torch.device('cpu')
  return cfg


# ---------------------------------------------------------------------------------------------------
def test_sde_bit_identical(st):
  g = load('sde.npz')
  x, t = torch.from_numpy(g['x']), torch.from_numpy(g['t'])
  cfgs = {'vp': st.configs.cifar10_ddpmpp_nll_st(), 've': st.configs.celebahq_uncsnpp_st(),
          'rve': st.configs.celeba_uncsnpp_st()}
  for name, cfg in cfgs.items():
    sde = st.sde_lib.get_sde(cfg, None)
    mean, std = sde.marginal_prob(x, t)
    drift, diff = sde.sde(x, t)
    for key, val in (('mean', mean), ('std', std), ('drift', drift), ('diffusion', diff), ('prior_logp', sde.prior_logp(x))):
      assert np.array_equal(val.numpy(), g[f'{name}.{key}']), f'{name}.{key}'
    if name != 'rve':
      f, G = sde.discretize(x, t)
      assert np.array_equal(f.numpy(), g[f'{name}.disc_f']) and np.array_equal(G.numpy(), g[f'{name}.disc_G'])
      assert np.array_equal(sde.normalizing_constant(1e-3).numpy(), g[f'{name}.Z'])
      assert np.array_equal(sde.antiderivative(t).numpy(), g[f'{name}.antiderivative'])
    for tm_name, t_min in (('eps', 1e-5), ('mid', 3e-3)):
      for imp in (True, False):
        with patched_rng(5):
          tt, Z = sde.get_diffusion_time(cfg, 6, torch.device('cpu'), t_min, importance_sampling=imp)
        assert np.array_equal(tt.numpy(), g[f'{name}.time.{tm_name}.{int(imp)}']), f'{name} time {tm_name} {imp}'
        assert float(Z) == float(g[f'{name}.timeZ.{tm_name}.{int(imp)}'])
    np.random.seed(11)
    assert np.array_equal(np.asarray([sde.get_t_min(cfg) for _ in range(4)]), g[f'{name}.t_min'])
  cfg = cfgs['vp']
  cfg.training.k = 2.0
  np.random.seed(11)
  sde = st.sde_lib.get_sde(cfg, None)
  assert np.array_equal(np.asarray([sde.get_t_min(cfg) for _ in range(4)]), g['vp.t_min.k2'])
  # the VE / RVE quirk: get_t_min ignores training.st, so the step always truncates at eps
  assert st.sde_lib.get_sde(cfgs['ve'], None).get_t_min(cfgs['ve']) == 1e-5
  assert st.sde_lib.get_sde(cfgs['rve'], None).get_t_min(cfgs['rve']) == 1e-5


def test_upfirdn2d_checker_and_restatement(ref_lib):
  g = load('ops.npz')
  x = torch.from_numpy(g['x'])
  N, C, H, W = x.shape
  for name in ('down', 'up', 'pre', 'crop', 'odd'):
    k = torch.from_numpy(g[f'{name}.k'])
    up, down, p0, p1 = (int(v) for v in g[f'{name}.args'])
    want, go, gx_want = g[f'{name}.y'], torch.from_numpy(g[f'{name}.go']), g[f'{name}.gx']
    # torch restatement
    y = ref_torch.upfirdn2d_ref(x, k, up=up, down=down, pad=(p0, p1))
    assert np.allclose(y.numpy(), want, atol=1e-6), name
    # C checker, forward
    kh, kw = k.shape
    out = torch.zeros(want.shape)
    call(ref_lib, 'upfirdn2d_f32', x.contiguous(), k.contiguous(), out, N * C, H, W, 1, kh, kw, up, up, down, down, p0, p1, p0, p1)
    assert np.allclose(out.numpy(), want, atol=1e-6), name
    # C checker, backward = same operator with flipped taps, up<->down and g_pad (op/upfirdn2d.py:111-114)
    oh, ow = want.shape[2], want.shape[3]
    gp0x, gp0y = kw - p0 - 1, kh - p0 - 1
    gp1x = W * up - ow * down + p0 - up + 1
    gp1y = H * up - oh * down + p0 - up + 1
    gx = torch.zeros(N, C, H, W)
    call(ref_lib, 'upfirdn2d_f32', go.contiguous(), torch.flip(k, [0, 1]).contiguous(), gx, N * C, oh, ow, 1, kh, kw,
         down, down, up, up, gp0x, gp1x, gp0y, gp1y)
    assert np.allclose(gx.numpy(), gx_want, atol=1e-6), name + ' backward'
  assert np.allclose(ref_torch.upsample_2d(x, (1, 3, 3, 1)).numpy(), g['uds.up'], atol=1e-6)
  assert np.allclose(ref_torch.downsample_2d(x, (1, 3, 3, 1)).numpy(), g['uds.down'], atol=1e-6)
  assert np.allclose(ref_torch.conv_downsample_2d(x, torch.from_numpy(g['uds.w']), (1, 3, 3, 1)).numpy(), g['uds.conv_down'], atol=1e-5)
  assert np.array_equal(ref_torch.naive_upsample_2d(x).numpy(), g['uds.naive_up'])
  assert np.allclose(ref_torch.naive_downsample_2d(x).numpy(), g['uds.naive_down'], atol=1e-7)
  # C checker of the naive resamplers and of fused_bias_act
  up_o, dn_o = torch.zeros(N, C, 2 * H, 2 * W), torch.zeros(N, C, H // 2, W // 2)
  call(ref_lib, 'resample_naive_f32', x.contiguous(), up_o, N * C, H, W, 0, 1.0, 0.0)
  call(ref_lib, 'resample_naive_f32', x.contiguous(), dn_o, N * C, H, W, 1, 1.0, 0.0)
  assert np.array_equal(up_o.numpy(), g['uds.naive_up']) and np.allclose(dn_o.numpy(), g['uds.naive_down'], atol=1e-7)
  o = torch.zeros_like(x)
  call(ref_lib, 'fused_bias_act_f32', x.contiguous(), torch.from_numpy(g['flr.bias']), None, o, x.numel(), H * W, C, 3, 0,
       0.2, float(2 ** 0.5))
  assert np.allclose(o.numpy(), g['flr.y'], atol=1e-6)


def _ref_model(st, family, g):
  cfg = tiny_cfg(st, family)
  sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith('sd.')}
  return cfg, sd, st.models.utils.DataParallel(ref_torch.RefNet(cfg, sd))


@pytest.mark.parametrize('family', ['vp', 'rve', 've'])
def test_refnet_matches_reference(st, family):
  g = load(f'model_{family}.npz')
  cfg, sd, ref = _ref_model(st, family, g)
  sde = st.sde_lib.get_sde(cfg, None)
  x, t, cond = (torch.from_numpy(g[k]) for k in ('x', 't', 'cond'))
  ref.eval()
  xr = x.clone().requires_grad_(True)
  y = ref(xr, cond)
  assert np.allclose(y.detach().numpy(), g['net'], rtol=0, atol=1e-6 * np.abs(g['net']).max())
  (y * torch.from_numpy(g['go'])).sum().backward()
  assert np.allclose(xr.grad.numpy(), g['gx'], rtol=0, atol=2e-6 * np.abs(g['gx']).max())
  grads = dict(ref.named_parameters())
  scale = float(np.sqrt(g['grad_sumsq']))
  for n in g['grad_names']:
    got = grads[str(n).replace('.', '__').replace('module__', 'module.', 1)].grad.numpy()
    assert np.allclose(got, g['grad.' + str(n)], rtol=0, atol=1e-6 * scale), n
  s = st.models.utils.get_score_fn(cfg, sde, ref, train=False, continuous=True)(x, t)
  assert np.allclose(s.detach().numpy(), g['score'], rtol=0, atol=1e-6 * np.abs(g['score']).max())


@pytest.mark.parametrize('family', ['vp', 'rve', 've'])
def test_step_fn_matches_reference(st, family):
  """losses.get_step_fn + get_optimizer + EMA (host code shared by product and oracle) vs the reference."""
  g = load(f'model_{family}.npz')
  cfg, sd, ref = _ref_model(st, family, g)
  sde = st.sde_lib.get_sde(cfg, None)
  opt = st.losses.get_optimizer(cfg, ref.parameters())
  ema = st.models.ema.ExponentialMovingAverage(ref.parameters(), decay=cfg.model.ema_rate)
  state = dict(optimizer=opt, model=ref, ema=ema, step=0)
  step_fn = st.losses.get_step_fn(cfg, sde, train=True, optimize_fn=st.losses.optimization_manager(cfg))
  for i in range(2):
    np.random.seed(7 + i)
    with patched_rng(50 + i):
      losses = step_fn(state, torch.from_numpy(g[f'step{i}.batch']))
    assert np.allclose(losses.numpy(), g[f'step{i}.loss'], rtol=2e-6, atol=0), f'step {i}'
  assert state['step'] == int(g['after.step'])
  params = dict(ref.named_parameters())
  for n in g['grad_names']:
    got = params[str(n).replace('.', '__').replace('module__', 'module.', 1)].detach().numpy()
    assert np.allclose(got, g['after.' + str(n)], rtol=0, atol=2e-7), n
  assert np.allclose(ema.shadow_params[0].numpy(), g['after.ema0'], rtol=0, atol=2e-7)
  assert np.allclose(ema.shadow_params[-1].numpy(), g['after.ema_last'], rtol=0, atol=2e-7)


@pytest.mark.parametrize('family', ['vp', 've'])
def test_pc_sampler_matches_reference(st, family):
  g = load(f'model_{family}.npz')
  cfg, sd, ref = _ref_model(st, family, g)
  if family == 'vp':
    cfg.sampling.method, cfg.sampling.predictor, cfg.sampling.corrector = 'pc', 'euler_maruyama', 'none'
  sde = st.sde_lib.get_sde(cfg, None)
  sde.N = 3
  shape = (2, 3, cfg.data.image_size, cfg.data.image_size)
  fn = st.sampling.get_sampling_fn(cfg, sde, shape, st.datasets.get_data_inverse_scaler(cfg), 1e-3)
  with patched_rng(11):
    xs, nfe = fn(ref)
  assert nfe == int(g['pc.nfe'])
  assert np.allclose(xs.numpy(), g['pc.samples'], rtol=0, atol=2e-6 * np.abs(g['pc.samples']).max())


def test_sampler_registry_matches_reference(st):
  """ancestral_sampling, ald, reverse_diffusion / langevin on VP, euler_maruyama on VE, and the sub-VP SDE: the host
  code shared by product and oracle (sampling.py, sde_lib.py) on RefNet against the reference's outputs."""
  import _model_cases as cases

  def make(family):
    cfg, sd, ref = _ref_model(st, family, load(f'model_{family}.npz'))
    return cfg, ref

  cases.golden_sampler_registry(st, make, 5e-6, bit_exact_sde=True)


def test_sampler_registry_on_checker(st, ref_lib):
  """Same, with the planned-graph engine on the C checker as the network."""
  import _model_cases as cases
  cases.golden_sampler_registry_product(st, ref_lib)


@pytest.mark.parametrize('family', ['vp', 'rve', 've'])
def test_engine_on_checker_matches_reference(st, ref_lib, family):
  """Host logic of the product path (graph lowering, flat parameters, backward planning) on the C checker."""
  g = load(f'model_{family}.npz')
  cfg = tiny_cfg(st, family)
  net = st.models.ncsnpp.NCSNpp(cfg, None)
  net.load_state_dict({k[3 + len('module.'):]: torch.from_numpy(v) for k, v in g.items() if k.startswith('sd.')})
  net.set_backend(ref_lib)
  net.eval()
  x, cond = torch.from_numpy(g['x']), torch.from_numpy(g['cond'])
  xr = x.clone().requires_grad_(True)
  y = net(xr, cond)
  assert np.allclose(y.detach().numpy(), g['net'], rtol=0, atol=2e-5 * np.abs(g['net']).max())
  (y * torch.from_numpy(g['go'])).sum().backward()
  assert np.allclose(xr.grad.numpy(), g['gx'], rtol=0, atol=2e-5 * np.abs(g['gx']).max())
  grads = dict(net.named_parameters())
  scale = float(np.sqrt(g['grad_sumsq']))
  for n in g['grad_names']:
    assert np.allclose(grads[str(n)[len('module.'):]].grad.numpy(), g['grad.' + str(n)], rtol=0, atol=2e-5 * scale), n


def test_state_dict_layout_matches_reference(st):
  """Key names, order and shapes of the product model == the reference's (checkpoint compatibility)."""
  for family in ('vp', 'rve', 've'):
    g = load(f'model_{family}.npz')
    want = [(k[3:], v.shape) for k, v in g.items() if k.startswith('sd.')]
    cfg = tiny_cfg(st, family)
    model = st.models.utils.DataParallel(st.models.ncsnpp.NCSNpp(cfg, None))
    got = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    assert got == [(k, tuple(s)) for k, s in want]
    spec = ref_torch.build_spec(cfg)
    assert len(spec) == len(model.module.all_modules)


@pytest.mark.parametrize('family', ['vp', 've'])
def test_likelihood_matches_reference(st, family):
  """likelihood.py (host arithmetic around score_fn, device-resident RK45) on the oracle network == the reference's
  likelihood.py on the reference network: residual terms, soft-truncation NELBO, ODE NLL incl. the solver's nfev."""
  import _model_cases as cases
  g = load(f'model_{family}.npz')
  cfg, sd, ref = _ref_model(st, family, g)
  cases.golden_likelihood(st, ref, cfg, family, tol=2e-5)
