"""GPU parity tests at the model level: NCSNpp on the HIP engine (libstk.so, gfx950) against the oracle
RefNet on the host, through the reference's own interfaces (models.utils, losses.get_step_fn,
sampling.get_sampling_fn)."""
import pytest

import _model_cases as cases

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('family', ['vp', 'rve', 've'])
def test_forward_backward(st, hip_lib, family):
  cases.forward_backward(st, hip_lib, family)


@pytest.mark.parametrize('family', ['vp', 'rve', 've'])
def test_score_fn(st, hip_lib, family):
  cases.score_fn_parity(st, hip_lib, family)


@pytest.mark.parametrize('family', ['vp', 'rve', 've'])
def test_train_steps(st, hip_lib, family):
  cases.train_steps(st, hip_lib, family)


def test_train_steps_micro_batches(st, hip_lib):
  cases.train_steps(st, hip_lib, 'vp', steps=2, num_micro_batch=2)


def test_train_steps_mixed(st, hip_lib):
  cases.train_steps(st, hip_lib, 'vp', steps=2, mixed=True)


def test_dropout_consistency(st, hip_lib):
  cases.dropout_consistency(st, hip_lib)


@pytest.mark.parametrize('family', ['vp', 've'])
def test_pc_sampler(st, hip_lib, family):
  cases.pc_sampler_steps(st, hip_lib, family)


def test_ode_sampler(st, hip_lib):
  cases.ode_sampler(st, hip_lib)


def test_checkpoint_roundtrip(st, hip_lib, tmp_path):
  cases.checkpoint_roundtrip(st, hip_lib, tmp_path)


@pytest.mark.parametrize('family', ['vp', 'rve', 've'])
def test_golden_fixtures(st, hip_lib, family):
  cases.golden_forward_backward(st, hip_lib, family)


@pytest.mark.parametrize('family', ['vp', 've'])
def test_golden_likelihood(st, hip_lib, family):
  cases.golden_likelihood_product(st, hip_lib, family)


def test_product_fails_loudly_without_gpu_tensors(st, hip_lib):
  """The HIP backend refuses CPU tensors instead of falling back."""
  import torch
  cfg = st.configs.tiny(st.configs.cifar10_ddpmpp_nll_st())
  cfg.device = torch.device('cpu')
  net = st.models.ncsnpp.NCSNpp(cfg, None)
  with pytest.raises(RuntimeError, match='no CPU'):
    net(torch.zeros(1, 3, 16, 16), torch.zeros(1))
  with pytest.raises(RuntimeError, match='HIP'):
    st.op.upfirdn2d(torch.zeros(1, 1, 4, 4), torch.ones(2, 2))
