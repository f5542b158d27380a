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
