"""Host-logic tests on CPU: the planned-graph engine, FusedAdam, the fused EMA, the samplers and the
checkpoint format, executed with the oracle's C restatement INJECTED as the backend (test-only
dependency injection -- the product never selects it) and compared with the oracle RefNet."""
import pytest

import _model_cases as cases
from _model_util import build_pair, tiny_config
cases.build_pair, cases.tiny_config = build_pair, tiny_config


@pytest.mark.parametrize('family', ['vp', 'rve', 've'])
def test_forward_backward(st, ref_lib, family):
  cases.forward_backward(st, ref_lib, family)


@pytest.mark.parametrize('family', ['vp', 'rve', 've'])
def test_score_fn(st, ref_lib, family):
  cases.score_fn_parity(st, ref_lib, family)


@pytest.mark.parametrize('family', ['vp', 'rve', 've'])
def test_train_steps(st, ref_lib, family):
  cases.train_steps(st, ref_lib, family)


def test_loss_curve_on_the_checker(st, ref_lib):
  """The 100-step trajectory test of the GPU suite (tests/test_gpu_model.py::test_loss_curve_100_steps), 40 steps on the checker."""
  out = cases.loss_curve(st, ref_lib, 'vp', steps=40)
  print(out)


@pytest.mark.parametrize('family', ['vp_elu', 'vp_relu', 'vp_lrelu', 'vp_ff', 've_cat'])
def test_other_activations(st, ref_lib, family):
  """config.model.nonlinearity = elu / relu / lrelu on the checker: engine graph vs RefNet."""
  cases.forward_backward(st, ref_lib, family)
  cases.train_steps(st, ref_lib, family, steps=2)


def test_score_matching_loss_on_large_samples(st, ref_lib):
  cases.score_matching_pieces(st, ref_lib)


def test_train_steps_micro_batches(st, ref_lib):
  cases.train_steps(st, ref_lib, 'vp', steps=2, num_micro_batch=2)


def test_train_steps_mixed(st, ref_lib):
  cases.train_steps(st, ref_lib, 'vp', steps=2, mixed=True)


def test_dropout_consistency(st, ref_lib):
  cases.dropout_consistency(st, ref_lib)


@pytest.mark.parametrize('family', ['vp', 've'])
def test_pc_sampler(st, ref_lib, family):
  cases.pc_sampler_steps(st, ref_lib, family)


@pytest.mark.parametrize('family', ['vp', 've'])
def test_pc_sampler_lockstep_on_the_checker(st, ref_lib, family):
  """The harness of the GPU suite's full-length trajectory test (tests/_model_cases.pc_sampler_full_length) on a 60-point
  grid: lockstep states, final samples, and equality with the sampler's own loop."""
  cases.pc_sampler_full_length(st, ref_lib, family, N=60)


def test_ode_sampler(st, ref_lib):
  cases.ode_sampler(st, ref_lib)


def test_checkpoint_roundtrip(st, ref_lib, tmp_path):
  cases.checkpoint_roundtrip(st, ref_lib, tmp_path)


def test_device_rk45_matches_scipy():
  """engine/rk45.py restates scipy's RK45 (the solver of the reference's ODE sampler, sampling.py:479) on torch
  tensors: same number of function evaluations and the same final state on a stiff-ish nonlinear system."""
  import importlib
  import numpy as np
  import torch
  from scipy import integrate
  rk = importlib.import_module('soft-truncation_amd.engine.rk45')
  rng = np.random.default_rng(0)
  n = 300
  A = rng.standard_normal((n, n)) / np.sqrt(n) - 0.5 * np.eye(n)
  At = torch.from_numpy(A)
  y0 = rng.standard_normal(n)
  for span, rtol, atol in (((1.0, 1e-3), 1e-5, 1e-5), ((0.0, 2.0), 1e-3, 1e-6), ((1.0, 1e-5), 1e-7, 1e-9)):
    sol = integrate.solve_ivp(lambda t, y: A @ np.tanh(y) * (1 + 0.5 * np.sin(3 * t)), span, y0, rtol=rtol, atol=atol,
                              method='RK45')
    y, nfev = rk.solve_ivp_rk45(lambda t, v: At @ torch.tanh(v) * (1 + 0.5 * np.sin(3 * t)), span, torch.from_numpy(y0),
                                rtol=rtol, atol=atol)
    assert nfev == sol.nfev
    assert np.abs(sol.y[:, -1] - y.numpy()).max() < 1e-11


def test_sample_postprocessing_on_checker(st, ref_lib, tmp_path):
  """sampling_lib: uint8 NHWC conversion == the reference's numpy expression (sampling_lib.py:43), grid layout ==
  torchvision.make_grid(nrow, padding=2) restated, npz round trip."""
  import numpy as np
  import torch
  g = torch.Generator().manual_seed(3)
  x = torch.rand(10, 3, 8, 8, generator=g) * 1.2 - 0.1            # some values outside [0, 1]
  got = st.sampling_lib.samples_to_uint8(x, backend=ref_lib)
  want = np.clip(x.permute(0, 2, 3, 1).numpy() * 255., 0, 255).astype(np.uint8)
  assert got.dtype == np.uint8 and got.shape == (10, 8, 8, 3)
  assert np.array_equal(got, want)
  grid = st.sampling_lib.make_grid_uint8(want, nrow=3, padding=2)
  assert grid.shape == (4 * 10 + 2, 3 * 10 + 2, 3)
  assert np.array_equal(grid[2:10, 2:10], want[0]) and np.array_equal(grid[12:20, 22:30], want[5])
  assert not grid[:2].any() and not grid[:, :2].any() and not grid[32:40, 12:].any()   # padding and the empty cells


def test_device_batch_on_checker(st, ref_lib):
  """datasets.device_batch == convert_image_dtype + flip + (255 x + u) / 256 + scaler (datasets.py:313-324,
  run_lib.py:72-75): exact without the random parts; with them, every image is the original or its mirror and the
  dequantisation noise is U[0,1) per element."""
  import numpy as np
  import torch
  g = torch.Generator().manual_seed(11)
  img = torch.randint(0, 256, (64, 8, 8, 3), generator=g, dtype=torch.uint8)
  cfg = st.configs.get_config('cifar10_ddpmpp_nll_st')          # centred, random_flip, no dequantisation
  plain = img.permute(0, 3, 1, 2).float() * (1.0 / 255.0)
  out = st.datasets.device_batch(cfg, img, seed=5, evaluation=True, backend=ref_lib)
  assert torch.equal(out, plain * 2. - 1.)
  out = st.datasets.device_batch(cfg, img, seed=5, backend=ref_lib)
  same = (out == plain * 2. - 1.).flatten(1).all(1)
  mirrored = (out == (plain * 2. - 1.).flip(3)).flatten(1).all(1)
  assert bool((same | mirrored).all()) and 10 < int(mirrored.sum()) < 54          # about half of the images flip
  cfg.data.dequantization, cfg.data.random_flip, cfg.data.centered = 'uniform', False, False
  out = st.datasets.device_batch(cfg, img, seed=9, backend=ref_lib)
  u = out * 256. - 255. * plain
  assert float(u.min()) > -1e-4 and float(u.max()) < 1. + 1e-4 and abs(float(u.mean()) - 0.5) < 0.02
  assert not torch.equal(out, st.datasets.device_batch(cfg, img, seed=10, backend=ref_lib))


@pytest.mark.parametrize('family', ['vp'])       # 've' (900 network evaluations on the plain-C checker) runs on the GPU only
def test_likelihood_engine_on_checker(st, ref_lib, family):
  """likelihood.py through the planned-graph engine (checker backend): exercises the engine's input-gradient path
  under torch.autograd.grad against the reference's fixtures."""
  cases.golden_likelihood_product(st, ref_lib, family)


def test_planes_plumbing_on_checker(st, ref_lib):
  """The 'wide' family (96 / 192 channels) is the smallest net whose convolutions qualify for the pre-split operand
  path (include/stk.h "Planes"): GroupNorm writes bound record + planes, the 3x3 convs and attention's q / k / v
  projections read them, data gradients split dy into the context's scratch.  Same answers as the oracle RefNet,
  and the plan must actually contain planes."""
  import torch
  from importlib import import_module
  G = import_module('soft-truncation_amd.engine.graph')
  cases.forward_backward(st, ref_lib, 'wide', B=2)
  cfg, cfg_cpu, sde, model, ref = cases.build_pair(st, cases.tiny_config(st, 'wide'), ref_lib)
  x = torch.randn(2, 3, 16, 16)
  model.eval()
  model(x.requires_grad_(True), torch.rand(2) * 999).sum().backward()
  progs = list(model.module.engine().programs.values())
  convs = [op for pr in progs for op in pr.graph.ops if isinstance(op, G.Conv)]
  assert any(op.pl_fwd and op.KH == 3 for op in convs) and any(op.pl_fwd and op.KH == 1 for op in convs)
  assert any(op.pl_dgrad for op in convs)
  assert all(pr.graph.pl_bytes > 0 and pr.graph.dypl_bytes > 0 for pr in progs)
  gns = [op for pr in progs for op in pr.graph.ops if isinstance(op, G.GroupNormAct)]
  assert any(op.y.pl_maker is op for op in gns)
  # Conv_0 of a ResnetBlock: its output gradient is GroupNorm_1's dx1, whose backward leaves the bias / time-embedding
  # sums and the |dy| record behind (stk_gn_bwd_out_f32); the plan ends with the op that zeroes those records
  served = [op for op in convs if op.dy_prod is not None]
  assert served and all(op.dy_prod.dy_cons is op and op.bsum_index is not None for op in served)
  assert any(op.temb is not None for op in served)                 # Conv_0 -> GroupNorm_1
  assert any(op.temb is None and op.res is not None for op in served)   # a block's output -> the next block's GroupNorm_0
  assert any(op.dy_peer is not None for op in served)              # ... with a shortcut peer sharing record and fold entry
  # identity skips: the block's first GroupNorm adds d(out) / sqrt 2 into d(x) (no pass of Conv_1's backward over d(res))
  assert any(op.res_via is not None and op.res_via.add_from is op for op in convs)
  assert all(isinstance(pr.graph.ops[-1], G.ZeroRecords) for pr in progs)


def test_dy_producer_switch_off_gives_same_gradients(st, ref_lib, monkeypatch):
  """STK_DY_PRODUCER=0 STK_RES_VIA=0 (every convolution sums and measures its own dy and writes d(res)) against the default plan, on the checker: the
  same parameter gradients (bias gradients through the batched fold, time-embedding gradients written by the GroupNorm
  backward) to rounding."""
  import torch
  cfg, cfg_cpu, sde, model, ref = cases.build_pair(st, cases.tiny_config(st, 'wide'), ref_lib)
  model.train()
  x, t = torch.randn(2, 3, 16, 16), torch.rand(2) * 999

  def grads():
    model.zero_grad()
    model(x, t).square().sum().backward()
    return {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}

  torch.manual_seed(0)
  g1 = grads()
  monkeypatch.setenv('STK_DY_PRODUCER', '0')
  monkeypatch.setenv('STK_RES_VIA', '0')
  model.module.engine().programs.clear()
  torch.manual_seed(0)
  g0 = grads()
  assert g0.keys() == g1.keys()
  top = max(v.abs().max().item() for v in g0.values())
  for k in g0:    # (floor: gradients that are zero analytically -- attention's key bias -- are rounding noise in both plans)
    assert (g1[k] - g0[k]).abs().max().item() <= 2e-5 * max(g0[k].abs().max().item(), 1e-4 * top), k


def test_input_gradient_only_backward_leaves_parameter_gradients_alone(st, ref_lib):
  """torch.autograd.grad(out, x) -- the Hutchinson divergence of likelihood.py:38-47 -- must return the same input
  gradient as a full backward and must neither compute nor accumulate parameter gradients (in the reference autograd
  only walks the branches that were asked for)."""
  import torch
  cfg, cfg_cpu, sde, model, ref = cases.build_pair(st, cases.tiny_config(st, 'wide'), ref_lib)
  model.eval()
  x, t = torch.randn(2, 3, 16, 16), torch.rand(2) * 999
  xa = x.clone().requires_grad_(True)
  model.zero_grad()
  model(xa, t).square().sum().backward()
  gx_full = xa.grad.clone()
  flat = model.module.engine().flat
  assert float(flat.grad.abs().max()) > 0
  flat.grad.zero_()
  xb = x.clone().requires_grad_(True)
  gx, = torch.autograd.grad(model(xb, t).square().sum(), xb)
  assert float(flat.grad.abs().max()) == 0.0            # untouched
  assert torch.equal(gx, gx_full)
  # and a full backward afterwards still produces them
  xc = x.clone().requires_grad_(True)
  model(xc, t).square().sum().backward()
  assert float(flat.grad.abs().max()) > 0


def test_planes_switch_off_gives_same_answers(st, ref_lib, monkeypatch):
  """STK_PLANES=0 (every convolution on fp32 operands) against the default plan, on the checker: the restatement
  decodes planes exactly, so forward values agree to the split's 2^-22."""
  import torch
  cfg, cfg_cpu, sde, model, ref = cases.build_pair(st, cases.tiny_config(st, 'wide'), ref_lib)
  model.eval()
  x, t = torch.randn(2, 3, 16, 16), torch.rand(2) * 999
  with torch.no_grad():
    y1 = model(x, t)
  monkeypatch.setenv('STK_PLANES', '0')
  model.module.engine().programs.clear()
  with torch.no_grad():
    y0 = model(x, t)
  assert all(pr.graph.pl_bytes == 0 for pr in model.module.engine().programs.values())
  assert (y1 - y0).abs().max().item() <= 2e-6 * y0.abs().max().item()


def test_linear_data_gradient_k_split(st, ref_lib, monkeypatch):
  """The stacked time-embedding projection (46 Dense_0 layers, fout = 9984 at full size) computes its data gradient as a
  batched GEMM over K slices plus a ones-row GEMM that sums the slabs in slice order (engine/graph.py Linear).  Forced on
  for the tiny net here (slices of 16 columns); the full-size GPU parity tests run it at its real size."""
  from importlib import import_module
  G = import_module('soft-truncation_amd.engine.graph')
  monkeypatch.setattr(G.Linear, 'KSLICE', 16)
  monkeypatch.setattr(G.Linear, 'KSPLIT_MIN', 32)
  cases.forward_backward(st, ref_lib, 'vp')
  cfg, cfg_cpu, sde, model, ref = cases.build_pair(st, cases.tiny_config(st, 'vp'), ref_lib)
  import torch
  model(torch.randn(2, 3, 16, 16), torch.rand(2) * 999).sum().backward()
  lins = [op for pr in model.module.engine().programs.values() for op in pr.graph.ops if isinstance(op, G.Linear)]
  assert any(op.ksplit > 1 for op in lins)


def test_qkv_projection_is_planned_stacked(st, ref_lib):
  """AttnBlockpp's q / k / v NIN layers share their input: the flat layout interleaves their [in, out] weights into one
  [in, 3 out] matrix (engine/flat.py 'cols' group) and the plan holds ONE 1x1 convolution with 3 C output channels whose
  channel slices the attention op reads through a batch stride.  The parameters stay separate nn.Parameters (strided
  views) with the reference's names, values and gradients -- checked against RefNet by forward_backward."""
  import torch
  from importlib import import_module
  G = import_module('soft-truncation_amd.engine.graph')
  cases.forward_backward(st, ref_lib, 'vp')
  cfg, cfg_cpu, sde, model, ref = cases.build_pair(st, cases.tiny_config(st, 'vp'), ref_lib)
  model(torch.randn(2, 3, 16, 16), torch.rand(2) * 999).sum().backward()
  ops = [op for pr in model.module.engine().programs.values() for op in pr.graph.ops]
  attn = [op for op in ops if isinstance(op, G.AttentionCore)]
  assert attn and all(op.qkv is not None for op in attn)
  net = model.module
  blk = next(m for m in net.modules() if type(m).__name__ == 'AttnBlockpp')
  assert not blk.NIN_1.W.is_contiguous() and blk.NIN_1.W.grad is not None and blk.NIN_1.W.grad.abs().sum() > 0
  sd = model.state_dict()
  key = next(k for k in sd if k.endswith('NIN_1.W'))
  assert sd[key].shape == blk.NIN_1.W.shape


def test_shortcut_convolution_shares_the_block_output_gradient(st, ref_lib, monkeypatch):
  """A ResnetBlock with a shortcut convolution is planned so that Conv_2 (1x1 on the block input) differentiates from
  Conv_1's output gradient directly (engine/graph.py Graph._plan_shared_dy): d(Conv_2 out) = dy / sqrt 2 is never
  written, measured or summed on its own.  Same gradients as the plain plan (STK_SHARED_DY=0) and as RefNet."""
  import torch
  from importlib import import_module
  G = import_module('soft-truncation_amd.engine.graph')
  cases.forward_backward(st, ref_lib, 'wide', B=2)                       # against RefNet, with the shared plan
  cfg, cfg_cpu, sde, model, ref = cases.build_pair(st, cases.tiny_config(st, 'wide'), ref_lib)
  x, t = torch.randn(2, 3, 16, 16), torch.rand(2) * 999
  model(x, t).square().sum().backward()
  convs = [op for pr in model.module.engine().programs.values() for op in pr.graph.ops if isinstance(op, G.Conv)]
  pairs = [(op, op.dy_peer) for op in convs if op.dy_peer is not None]
  assert pairs and all(p.dy_from is c and p.KH == 1 and c.KH == 3 for c, p in pairs)
  # round 5: the shortcut's data gradient reads the dy planes its peer's split pass made (Conv.peer_planes; 'scratch' on one stream)
  assert any(p.peer_planes == 'scratch' for c, p in pairs), [p.peer_planes for c, p in pairs]
  shared = [p.grad.clone() for p in model.parameters()]
  monkeypatch.setenv('STK_SHARED_DY', '0')
  model.module.engine().programs.clear()
  for p in model.parameters():
    p.grad.zero_()
  model(x, t).square().sum().backward()
  convs = [op for pr in model.module.engine().programs.values() for op in pr.graph.ops if isinstance(op, G.Conv)]
  assert all(op.dy_peer is None and op.dy_from is None for op in convs)
  # (per-tensor scale with a floor: some gradients are zero in exact arithmetic -- the key bias of attention -- and hold
  # only round-off)
  top = max(p.grad.abs().max().item() for p in model.parameters())
  for a, p in zip(shared, model.parameters()):
    assert (a - p.grad).abs().max().item() <= 1e-5 * max(p.grad.abs().max().item(), 1e-4 * top)


def test_rebinding_one_middle_parameter_is_noticed_on_the_next_call(st, ref_lib):
  """engine/flat.py: every network evaluation and every zero_grad() checks ALL parameters against the flat buffers (raw
  addresses), so a single `p.data = ...` / `p.grad = None` on a parameter in the middle of the model -- legal against the
  reference -- takes effect on the very next call instead of training on a stale buffer."""
  import torch
  cfg, cfg_cpu, sde, model, ref = build_pair(st, tiny_config(st, 'vp'), ref_lib)
  x = torch.randn(2, 3, 16, 16, generator=torch.Generator().manual_seed(1))
  t = torch.rand(2, generator=torch.Generator().manual_seed(2)) * 999
  ex = model.module.engine()
  with torch.no_grad():
    y0 = model(x, t).clone()
  flat0 = ex.flat
  params = [p for p in model.parameters() if p.requires_grad]
  mid = params[len(params) // 2]
  assert mid is not flat0.module_order[0] and mid is not flat0.module_order[-1]
  # 1) rebinding the data of ONE parameter: the next evaluation must use the new values (here: a new layout is built)
  mid.data = mid.data.clone() * 3.0 + 0.5
  with torch.no_grad():
    y1 = model(x, t)
  assert ex.flat is not flat0 and ex.flat.is_bound()
  assert (y1 - y0).abs().max().item() > 1e-4
  ref_p = dict(ref.named_parameters())
  name = [k for k, p in model.named_parameters() if p is mid][0]
  with torch.no_grad():
    ref_p[name.replace('.', '__').replace('module__', 'module.', 1)].copy_(mid.data)
    yr = ref(x, t)
  assert (y1 - yr).abs().max().item() <= 2e-5 * max(yr.abs().max().item(), 1.0)
  # 2) dropping the gradient of ONE parameter: the next backward still lands in the flat gradient buffer, and the
  #    parameter sees it
  flat1 = ex.flat
  model(x, t).sum().backward()
  mid.grad = None
  assert not flat1.is_bound()
  model(x, t).sum().backward()
  assert ex.flat is flat1 and flat1.is_bound()
  assert mid.grad is not None and mid.grad.data_ptr() == flat1.view_of(flat1.grad, mid).data_ptr()
  assert mid.grad.abs().max().item() > 0
  # 3) the optimizer's zero_grad notices it as well
  opt = st.losses.get_optimizer(cfg, model.parameters())
  opt._backend = ref_lib
  mid.grad = None
  opt.zero_grad()
  assert flat1.is_bound() and float(flat1.grad.abs().max()) == 0.0


def test_flat_of_rejects_partial_and_foreign_lists(st, ref_lib):
  import importlib
  import torch
  flat_mod = importlib.import_module('soft-truncation_amd.engine.flat')
  cfg, cfg_cpu, sde, model, ref = build_pair(st, tiny_config(st, 'vp'), ref_lib)
  params = list(model.parameters())
  assert flat_mod.flat_of(params) is model.module.engine().flat
  assert flat_mod.flat_of(params[:-3]) is None                                   # a partial list
  assert flat_mod.flat_of(params[:-1] + [torch.nn.Parameter(torch.zeros(3))]) is None   # a plain tensor mixed in
  cfg2, _, _, other, _ = build_pair(st, tiny_config(st, 'vp'), ref_lib, seed=1)
  mixed = list(params)
  mixed[5] = list(other.parameters())[5]
  assert flat_mod.flat_of(mixed) is None                                         # a parameter of another buffer
  with pytest.raises(Exception):
    st.engine.optim.FusedAdam(mixed, backend=ref_lib)._bind()


def test_fused_loss_path_on_the_checker(st, ref_lib):
  """losses.get_sde_loss_fn's three-kernel path on the CPU checker against the torch expressions (VP and VE)."""
  import _model_cases as cases
  for family in ('vp', 've'):
    cases.fused_loss_matches_torch(st, ref_lib, family)


def test_train_steps_amsgrad_on_the_checker(st, ref_lib):
  """optim.amsgrad=True (losses.py:33): FusedAdam keeps the running maximum of the second moment like torch's Adam, and
  publishes it under torch's state key."""
  import _model_cases as cases
  cases.train_steps(st, ref_lib, 'vp', steps=4, amsgrad=True)
  from _model_util import build_pair, make_state, tiny_config
  cfg = tiny_config(st, 'vp')
  cfg.optim.amsgrad = True
  cfg, cfg_cpu, sde, model, ref = build_pair(st, cfg, ref_lib)
  opt = st.losses.get_optimizer(cfg, model.parameters())
  opt._backend = ref_lib
  opt.zero_grad()
  opt.step()
  sd = opt.state_dict()
  assert all('max_exp_avg_sq' in v for v in sd['state'].values())


def test_raised_step_with_collectives_in_flight_poisons_the_exchange(st, ref_lib, monkeypatch):
  """engine/ddp.py: a step that raises while bucket all-reduces are in flight drops the handles without waiting (a wait could
  hang on peers that never issue the matching call) and marks the exchange unusable: the next arm_overlap / sync_gradients
  raises a clear error instead of issuing collectives the ranks no longer agree on."""
  from _model_util import build_pair, make_state, tiny_config
  ddp = st.engine.ddp
  cfg, cfg_cpu, sde, model, ref = build_pair(st, tiny_config(st, 'vp'), ref_lib)
  state = make_state(st, cfg, model)
  state['optimizer']._backend = ref_lib
  monkeypatch.setattr(ddp, 'is_distributed', lambda: True)
  monkeypatch.setattr(ddp, '_poisoned', None)
  x = ddp.arm_overlap(model)
  assert x is not None and model.module.engine().grad_hook is not None
  x.handles.append(object())                      # one bucket all-reduce "in flight"
  ddp.disarm_overlap(model, wait=False)           # what step_fn does when the step raises
  assert model.module.engine().grad_hook is None
  with pytest.raises(RuntimeError, match='gradient exchange unusable'):
    ddp.arm_overlap(model)
  with pytest.raises(RuntimeError, match='gradient exchange unusable'):
    ddp.sync_gradients(state['optimizer'])
  ddp.reset_poison()
  assert ddp.arm_overlap(model) is not None
  ddp.disarm_overlap(model, wait=False)           # nothing in flight: not poisoned
  assert ddp._poisoned is None


def test_conv_emitter_checks_the_weight_width(st, ref_lib):
  """Graph.conv refuses an input whose channel count differs from the weight's (the kernels never read the weight's width);
  NCSNpp refuses model.fourier_feature with non-RGB data, where the reference's channels + 12 stem cannot match 5 x channels."""
  import torch
  from importlib import import_module
  from _model_util import tiny_config
  G = import_module('soft-truncation_amd.engine.graph')
  cfg = tiny_config(st, 'vp_ff')
  cfg.data.num_channels = 1
  with pytest.raises(ValueError, match='fourier_feature'):
    st.models.ncsnpp.NCSNpp(cfg, None)
  cfg = tiny_config(st, 'vp')
  net = st.models.ncsnpp.NCSNpp(cfg, None)
  net.set_backend(ref_lib)
  flat = net.engine().ensure_flat()
  g = G.Graph(flat, ref_lib)
  x = g.input('x', (2, 5, 16, 16))                # the stem expects 3 channels
  conv = next(m for m in net.all_modules if getattr(m, 'weight', None) is not None and m.weight.dim() == 4)
  with pytest.raises(RuntimeError, match='expected input with 3 channels'):
    g.conv(x, None, conv.weight, conv.bias, name='stem')


def test_dynamic_range_report_flags_faint_images(st, ref_lib):
  """Executor.dynamic_range_report: per-image maxima of the convolutions' fp32 output gradients (read back from the gradient arena);
  a batch whose loss gradient spans seven decades across images is reported as such and losses._warn_dynamic_range turns it into a
  warning, an even batch is not."""
  import torch
  import warnings
  cfg, cfg_cpu, sde, model, ref = cases.build_pair(st, cases.tiny_config(st, 'wide'), ref_lib)
  ex = model.module.engine()
  B = 4
  x, t = torch.randn(B, 3, 16, 16, generator=torch.Generator().manual_seed(1)), torch.rand(B) * 999
  go = torch.randn(B, 3, 16, 16, generator=torch.Generator().manual_seed(2))
  model(x, t).mul(go).sum().backward()
  rows = ex.dynamic_range_report()
  assert len(rows) >= 4 and rows[0][3] < 2.0, rows[:3]
  with warnings.catch_warnings():
    warnings.simplefilter('error')
    st.losses._warn_dynamic_range(model, 0)                     # no warning for an even batch
  scale = torch.tensor([1.0, 1e-7, 1.0, 1.0]).view(B, 1, 1, 1)
  model.zero_grad()
  model(x, t).mul(go * scale).sum().backward()
  rows = ex.dynamic_range_report()
  assert rows[0][3] > 6.0 and all(r[1] >= r[2] > 0 for r in rows), rows[:3]
  with pytest.warns(UserWarning, match='decades across the images'):
    st.losses._warn_dynamic_range(model, 0)
