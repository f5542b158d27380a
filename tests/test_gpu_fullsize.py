"""GPU parity at the BASELINE sizes: the full 61.8 M / 62.8 M / 65.6 M-parameter networks on the HIP engine against
the oracle RefNet (tests/_fullsize_cases.py).  BASELINE.json configs[0] is reproduced verbatim (batch 8, one training
step + one PC iteration); the 64x64 and 256x256 nets run forward + backward at a batch the host restatement finishes in
about a minute."""
import pytest

import _fullsize_cases as full

pytestmark = pytest.mark.gpu


def test_baseline_config0_ddpmpp_cifar10(st, hip_lib):
  out = full.baseline_config0(st, hip_lib)
  print('configs[0] parity:', {k: (f'{v:.2e}' if isinstance(v, float) else v) for k, v in out.items()})


def test_full_uncsnpp_celeba64(st, hip_lib):
  """UNCSN++ (RVE) CelebA 64x64 (BASELINE configs[2] net): FIR resampling through upfirdn2d, progressive_input
  'residual', Fourier embedding, scale_by_sigma; attention at 16x16 (T = 256)."""
  out = full.full_forward_backward(st, hip_lib, 'celeba_uncsnpp_st', B=4)
  print('UNCSN++ 64 parity:', {k: (f'{v:.2e}' if isinstance(v, float) else v) for k, v in out.items()})
  if not full.SHRINK and hip_lib.is_device:
    v = out['variants']
    assert v.get('conv3x3.fwd.x2', 0) > 0 and v.get('conv3x3.wgrad.x2', 0) > 0, v


def test_full_ncsnpp_celebahq256(st, hip_lib):
  """NCSN++ (VE) CelebA-HQ 256x256 (BASELINE configs[4] net), batch 1: seven levels, input_skip / output_skip
  pyramids (thin-side kernels), 18 + 18 FIR resamplings, few-tile / K-split convolutions at the 4x4 .. 16x16 levels."""
  out = full.full_forward_backward(st, hip_lib, 'celebahq_uncsnpp_st', B=1, shrink_kw=dict(ch_mult=(1, 1, 2)))
  print('NCSN++ 256 parity:', {k: (f'{v:.2e}' if isinstance(v, float) else v) for k, v in out.items()})
  if not full.SHRINK and hip_lib.is_device:
    v = out['variants']
    assert any(k.endswith('.thin') for k in v) and any(k.endswith('.x2') for k in v), v
