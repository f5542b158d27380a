"""Host-logic tests on CPU: the planned-graph engine, FusedAdam, the fused EMA, the samplers and the
checkpoint format, executed with the oracle's C restatement INJECTED as the backend (test-only
dependency injection -- the product never selects it) and compared with the oracle RefNet."""
import pytest

import _model_cases as cases


@pytest.mark.parametrize('family', ['vp', 'rve', 've'])
def test_forward_backward(st, ref_lib, family):
  cases.forward_backward(st, ref_lib, family)


@pytest.mark.parametrize('family', ['vp', 'rve', 've'])
def test_score_fn(st, ref_lib, family):
  cases.score_fn_parity(st, ref_lib, family)


@pytest.mark.parametrize('family', ['vp', 'rve', 've'])
def test_train_steps(st, ref_lib, family):
  cases.train_steps(st, ref_lib, family)


def test_train_steps_micro_batches(st, ref_lib):
  cases.train_steps(st, ref_lib, 'vp', steps=2, num_micro_batch=2)


def test_train_steps_mixed(st, ref_lib):
  cases.train_steps(st, ref_lib, 'vp', steps=2, mixed=True)


def test_dropout_consistency(st, ref_lib):
  cases.dropout_consistency(st, ref_lib)


@pytest.mark.parametrize('family', ['vp', 've'])
def test_pc_sampler(st, ref_lib, family):
  cases.pc_sampler_steps(st, ref_lib, family)


def test_ode_sampler(st, ref_lib):
  cases.ode_sampler(st, ref_lib)


def test_checkpoint_roundtrip(st, ref_lib, tmp_path):
  cases.checkpoint_roundtrip(st, ref_lib, tmp_path)
