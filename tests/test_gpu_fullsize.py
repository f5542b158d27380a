"""GPU parity at the BASELINE sizes: the full 61.8 M / 62.8 M / 65.6 M-parameter networks on the HIP engine against
the oracle RefNet (tests/_fullsize_cases.py).  BASELINE.json configs[0] is reproduced verbatim (batch 8, one training
step + one PC iteration); the 64x64 and 256x256 nets run forward + backward at a batch the host restatement finishes in
about a minute."""
import pytest

import _fullsize_cases as full

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _release_device_memory():
  """Every case here builds a full-size engine (tens of GB of arenas at batch 128): drop it before the next one starts --
  without this the 64x64 batch-128 case and the 256x256 case only fit together when the garbage collector happened to run."""
  yield
  import gc
  import torch
  gc.collect()
  if torch.cuda.is_available():
    torch.cuda.synchronize()
    torch.cuda.empty_cache()


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


@pytest.mark.parametrize('cfg_name', ['cifar10_ddpmpp_nll_st', 'imagenet32_ddpmpp_st'])
def test_benched_batch_128_equals_sixteen_batch_8_runs(st, hip_lib, cfg_name):
  """BASELINE configs[1] / configs[3] at the benched per-GPU batch (128): sample for sample equal to sixteen batch-8 runs,
  chunk 0 equal to the oracle, and the plan really is the benched one (plane GEMMs, K-split small maps, all four
  weight-gradient tilings, more than one slab)."""
  out = full.benched_batch_vs_chunks(st, hip_lib, cfg_name)
  print(cfg_name, 'batch-128 parity:', {k: (f'{v:.2e}' if isinstance(v, float) else v) for k, v in out.items()})


@pytest.mark.timeout(3000)
@pytest.mark.parametrize('cfg_name,B,chunk', [('celeba_uncsnpp_st', 128, 8), ('celebahq_uncsnpp_st', 4, 1)])
def test_benched_batch_of_the_other_baseline_nets(st, hip_lib, cfg_name, B, chunk):
  """BASELINE configs[2] (UNCSN++ 64x64, per-GPU batch 128) and configs[4] (NCSN++ 256x256, per-GPU batch 4) at the batch
  bench.py / the scaling run use: tiles, K splits, halo widths and slab counts depend on the batch, so the benched plan is
  compared sample for sample with chunked runs (chunk 0 against the oracle), as for the 32x32 nets above."""
  out = full.benched_batch_vs_chunks(st, hip_lib, cfg_name, B=B, chunk=chunk)
  print(cfg_name, f'batch-{B} parity:', {k: (f'{v:.2e}' if isinstance(v, float) else v) for k, v in out.items()})


def test_full_uncsnpp_celeba64_train_step(st, hip_lib):
  """BASELINE configs[2] net: two full `step_fn` calls at batch 2 (RVE loss, sum reduction, FIR resampling, Adam, EMA)."""
  out = full.full_train_step(st, hip_lib, 'celeba_uncsnpp_st', B=2)
  print('UNCSN++ 64 step parity:', {k: (f'{v:.2e}' if isinstance(v, float) else v) for k, v in out.items()})


def test_full_ncsnpp_celebahq256_pc_iteration(st, hip_lib):
  """BASELINE configs[4] sampler leg: one reverse_diffusion + langevin iteration of the 256x256 NCSN++ at batch 1."""
  out = full.full_pc_iteration(st, hip_lib, 'celebahq_uncsnpp_st', B=1, shrink_kw=dict(ch_mult=(1, 1, 2)))
  print('NCSN++ 256 PC iteration parity:', {k: (f'{v:.2e}' if isinstance(v, float) else v) for k, v in out.items()})
  assert out['predictor'] == 'ReverseDiffusionPredictor' and out['corrector'] == 'LangevinCorrector'


def test_benched_batch_128_two_streams_equal_one(st, hip_lib):
  """BASELINE configs[1] at batch 128: side-stream backward == one-stream backward, bit for bit (engine/executor.SideStream)."""
  out = full.full_two_streams(st, hip_lib, 'cifar10_ddpmpp_nll_st')
  print('two streams:', out)
