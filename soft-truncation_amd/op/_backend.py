"""Backend binding shared by the `op` functions: the HIP library, loaded on first use."""
from ..engine import lib as stk_lib

_backend = None


def get():
  global _backend
  if _backend is None:
    _backend = stk_lib.load()
  return _backend


def set_backend(backend):
  """Test hook: bind another implementation of include/stk.h (the oracle's CPU checker)."""
  global _backend
  _backend = backend


def check(t, lib):
  if lib.is_device != (t.device.type == 'cuda'):
    raise RuntimeError(f'tensor on {t.device} but backend is {lib.backend}: this op runs on the HIP '
                       f'kernels only (no CPU / PyTorch fallback)')
