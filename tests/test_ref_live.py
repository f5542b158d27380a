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
