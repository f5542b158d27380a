"""Checks against the LIVE reference (its Python CPU path imported from /root/reference through
oracle/refimport.py).  Only possible in the build container; skipped wherever the reference is absent
(it never travels to the GPU box).  The committed fixtures of tests/golden/ carry the same pinning."""
import numpy as np
import pytest
import torch

import refimport

pytestmark = pytest.mark.skipif(not refimport.available(), reason='/root/reference is not present on this machine')

FULL = {
  'cifar10_ddpmpp_nll_st': ('configs.vp.CIFAR10.ddpmpp_nll_st', 61804419),
  'celeba_uncsnpp_st': ('configs.ve.CELEBA.uncsnpp_st', 62758915),
  'celebahq_uncsnpp_st': ('configs.ve.celebahq.uncsnpp_st', 65574549),
}


def _flatten(d, prefix=''):
  out = {}
  for k, v in d.items():
    if hasattr(v, 'items'):
      out.update(_flatten(v, prefix + k + '.'))
    else:
      out[prefix + k] = v
  return out


@pytest.mark.parametrize('name', sorted(FULL))
def test_config_values_match_reference(st, name):
  """Every value of the restated BASELINE configs equals the reference's get_config()."""
  ref = _flatten(refimport.get_config(FULL[name][0]))
  ours = _flatten(st.configs.get_config(name))
  ref.pop('device'); ours.pop('device')
  ref.pop('data.tfrecords_path', None)
  assert set(ref) == set(ours), (sorted(set(ref) - set(ours)), sorted(set(ours) - set(ref)))
  for k in ref:
    a, b = ref[k], ours[k]
    if isinstance(a, (tuple, list)):
      assert tuple(a) == tuple(b), k
    else:
      assert a == b, (k, a, b)


@pytest.mark.parametrize('name', sorted(FULL))
def test_full_size_state_dict_matches_reference(st, name):
  """Full-size models: parameter count, state_dict keys, order, shapes and dtypes (checkpoint compatibility)."""
  ns = refimport.load()
  rcfg = refimport.get_config(FULL[name][0])
  rmodel = ns.mutils.create_model(rcfg, ns.sde_lib.get_sde(rcfg, None))
  cfg = st.configs.get_config(name)
  cfg.device = torch.device('cpu')
  model = st.models.utils.DataParallel(st.models.ncsnpp.NCSNpp(cfg, None))
  assert sum(p.numel() for p in model.parameters()) == FULL[name][1] == sum(p.numel() for p in rmodel.parameters())
  a = [(k, tuple(v.shape), v.dtype) for k, v in rmodel.state_dict().items()]
  b = [(k, tuple(v.shape), v.dtype) for k, v in model.state_dict().items()]
  assert a == b
  assert [p.requires_grad for p in rmodel.parameters()] == [p.requires_grad for p in model.parameters()]


def test_init_statistics_match_reference(st):
  """Same initialisers: per-tensor std of a fresh product model == the reference's (same fan computations)."""
  ns = refimport.load()
  rcfg = refimport.get_config('configs.vp.CIFAR10.ddpmpp_nll_st')
  rcfg.model.nf, rcfg.model.num_res_blocks = 32, 1
  cfg = st.configs.cifar10_ddpmpp_nll_st()
  cfg.model.nf, cfg.model.num_res_blocks = 32, 1
  cfg.device = torch.device('cpu')
  torch.manual_seed(0)
  rmodel = ns.mutils.create_model(rcfg, ns.sde_lib.get_sde(rcfg, None))
  torch.manual_seed(0)
  model = st.models.utils.DataParallel(st.models.ncsnpp.NCSNpp(cfg, None))
  for (k, a), (_, b) in zip(rmodel.state_dict().items(), model.state_dict().items()):
    assert torch.equal(a, b), k     # same RNG consumption order and formulas -> identical tensors


@pytest.mark.parametrize('name', ['elu', 'relu', 'lrelu', 'swish+fourier_feature'])
def test_other_activations_match_reference(st, name):
  """config.model.nonlinearity != 'swish' (models/layers.py:29-41; no shipped config uses it): the LIVE reference NCSNpp on a
  fixture-size config against the oracle RefNet on the same state_dict -- pins the oracle the product's activation codes are
  tested against (tests/test_engine_cpu.py::test_other_activations, tests/test_gpu_model.py::test_other_activations)."""
  import ref_torch
  ns = refimport.load()
  rcfg = refimport.get_config('configs.vp.CIFAR10.ddpmpp_nll_st')
  cfg = st.configs.cifar10_ddpmpp_nll_st()
  for c in (rcfg, cfg):
    c.model.nf, c.model.ch_mult, c.model.num_res_blocks, c.model.attn_resolutions = 16, (1, 2), 1, (8,)
    c.model.dropout, c.data.image_size, c.model.nonlinearity = 0.0, 16, name.split('+')[0]
    c.model.fourier_feature = name.endswith('fourier_feature')
    c.device = torch.device('cpu')
  torch.manual_seed(0)
  rmodel = ns.mutils.create_model(rcfg, ns.sde_lib.get_sde(rcfg, None))
  g = torch.Generator().manual_seed(3)
  with torch.no_grad():
    for p in rmodel.parameters():
      if p.requires_grad:
        p.copy_(torch.randn(p.shape, generator=g) * 0.1)
  sd = {k: v.detach().clone() for k, v in rmodel.state_dict().items()}
  ref = ref_torch.RefNet(cfg, sd)
  x = torch.randn(3, 3, 16, 16, generator=g)
  t = torch.rand(3, generator=g) * 999
  rmodel.eval(); ref.eval()
  with torch.no_grad():
    want = rmodel(x, t)
    got = ref(x, t)
  err = float((got - want).abs().max() / want.abs().max())
  assert err <= 2e-6, (name, err)


def test_combine_cat_matches_reference(st):
  """progressive_combine = 'cat' (Combine, models/layerspp.py:57-72; channel doubling models/ncsnpp.py:183-184) on the NCSN++
  pyramid config: live reference vs RefNet on the same state_dict, and the product module list has the reference's keys / shapes."""
  import ref_torch
  ns = refimport.load()
  rcfg = refimport.get_config('configs.ve.celebahq.uncsnpp_st')
  cfg = st.configs.celebahq_uncsnpp_st()
  for c in (rcfg, cfg):
    c.model.nf, c.model.ch_mult, c.model.num_res_blocks, c.model.attn_resolutions = 16, (1, 1, 2), 1, (8,)
    c.model.dropout, c.data.image_size, c.model.progressive_combine = 0.0, 16, 'cat'
    c.device = torch.device('cpu')
  torch.manual_seed(0)
  rmodel = ns.mutils.create_model(rcfg, ns.sde_lib.get_sde(rcfg, None))
  torch.manual_seed(0)
  model = st.models.utils.DataParallel(st.models.ncsnpp.NCSNpp(cfg, None))
  rsd, sd = rmodel.state_dict(), model.state_dict()
  assert list(rsd) == list(sd) and all(rsd[k].shape == sd[k].shape for k in rsd)
  g = torch.Generator().manual_seed(3)
  with torch.no_grad():
    for p in rmodel.parameters():
      if p.requires_grad:
        p.copy_(torch.randn(p.shape, generator=g) * 0.1)
  ref = ref_torch.RefNet(cfg, {k: v.detach().clone() for k, v in rmodel.state_dict().items()})
  x = torch.rand(2, 3, 16, 16, generator=g)
  sig = torch.rand(2, generator=g) * 5 + 0.1
  rmodel.eval(); ref.eval()
  with torch.no_grad():
    want, got = rmodel(x, sig), ref(x, sig)
  err = float((got - want).abs().max() / want.abs().max())
  assert err <= 2e-6, err

