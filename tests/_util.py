"""Test helpers: call a C-ABI entry with torch tensors, on either backend."""
import numpy as np
import torch


def ptr(t):
  return None if t is None else t.data_ptr()


def call(lib, name, *args):
  """Tensors -> raw pointers; the stream is torch's current one (0 for the CPU checker)."""
  dev = None
  conv = []
  for a in args:
    if isinstance(a, torch.Tensor):
      assert a.is_contiguous() and a.dtype in (torch.float32, torch.int64, torch.uint8, torch.float64), a.dtype
      dev = a.device
      conv.append(a.data_ptr())
    else:
      conv.append(a)
  stream = 0
  if lib.is_device:
    stream = torch.cuda.current_stream().cuda_stream
  return getattr(lib, name)(*conv, stream)


def dev_of(lib):
  return torch.device('cuda:0') if lib.is_device else torch.device('cpu')


def rnd(*shape, seed=0, scale=1.0):
  g = torch.Generator().manual_seed(seed)
  return torch.randn(*shape, generator=g) * scale


def close(a, b, rtol=1e-5, atol=None, what=''):
  """max|a-b| <= rtol * max|b| + atol ; returns the error for reporting."""
  a = a.detach().cpu().double()
  b = b.detach().cpu().double()
  assert a.shape == b.shape, (a.shape, b.shape)
  scale = b.abs().max().item()
  if atol is None:
    atol = 1e-7
  err = (a - b).abs().max().item() if a.numel() else 0.0
  assert np.isfinite(err), f'{what}: non-finite result'
  assert err <= rtol * scale + atol, f'{what}: max abs err {err:.3e} vs scale {scale:.3e} (rtol {rtol})'
  return err
