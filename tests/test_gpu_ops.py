"""GPU execution of the reference's native-op surface through torch.autograd: `op.upfirdn2d` forward, backward and
double backward, `op.fused_leaky_relu` / `FusedLeakyReLU` forward, grad and grad-grad, and the tensor-level
up_or_down_sampling functions, on the HIP library (tests/_op_cases.py holds the cases and the oracle comparisons)."""
import pytest
import torch

import _op_cases as cases

pytestmark = pytest.mark.gpu


@pytest.fixture()
def dev(hip_lib):
  from importlib import import_module
  be = import_module('soft-truncation_amd.op._backend')
  saved = be._backend
  be.set_backend(hip_lib)       # STK_SELFCHECK runs bind the checker here; on a GPU box this is the HIP library
  yield torch.device('cuda:0') if hip_lib.is_device else torch.device('cpu')
  be._backend = saved


def test_upfirdn2d_forward_backward_double_backward(st, dev):
  cases.upfirdn2d_autograd(st, dev)
  cases.upfirdn2d_autograd(st, dev, shape=(4, 64, 32, 32))        # a feature-map-sized call (tiled kernel)


def test_upfirdn2d_against_reference_outputs(st, dev):
  cases.upfirdn2d_golden(st, dev)


def test_resampling_wrappers(st, dev):
  cases.resampling_wrappers_golden(st, dev)


def test_fused_leaky_relu_grad_and_gradgrad(st, dev):
  cases.fused_leaky_relu_autograd(st, dev)


def test_half_and_double_entry_points(st, dev):
  cases.other_dtypes(st, dev)
