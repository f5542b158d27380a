"""ctypes binding of the C ABI declared in include/stk.h.

``load()`` returns the product library (``csrc/libstk.so``, hand-written HIP for gfx950) and
raises :class:`StkMissingError` when it has not been built -- there is no CPU or PyTorch
fallback anywhere in the product path.  ``load_path()`` binds any library implementing the same
header; tests use it to inject the oracle's plain-C restatement as a *checker* backend for
host-logic tests on CPU tensors.
"""
import ctypes
import os
import threading
from ctypes import c_char_p, c_float, c_int, c_long, c_ulonglong, c_void_p

_PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRODUCT_LIB = os.path.join(_PKG_DIR, 'csrc', 'libstk.so')

P = c_void_p          # device (or host, for the oracle) pointer
S = c_void_p          # hipStream_t
I = c_int
L = c_long
F = c_float
U64 = c_ulonglong

# name -> argtypes, in the order of include/stk.h.  restype is int unless listed in _RESTYPE.
SIGNATURES = {
  'stk_strerror': [I],
  'stk_backend': [],
  'stk_version': [],
  'stk_upfirdn2d_f32': [P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, I, I, S],
  'stk_upfirdn2d_acc_f32': [P, P, P, F, I, I, I, I, I, I, I, I, I, I, I, I, I, I, S],
  'stk_fused_bias_act_f32': [P, P, P, P, L, I, I, I, I, F, F, S],
  'stk_fused_bias_act_f16': [P, P, P, P, L, I, I, I, I, F, F, S],
  'stk_fused_bias_act_f64': [P, P, P, P, L, I, I, I, I, F, F, S],
  'stk_upfirdn2d_f16': [P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, I, I, S],
  'stk_upfirdn2d_f64': [P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, I, I, S],
  'stk_gn_fwd_f32': [P, I, P, I, P, P, P, P, P, I, I, I, F, I, F, U64, P, P, S],
  'stk_gn_ws_bytes': [I, I, I, I],
  'stk_gn_bwd_f32': [P, P, I, P, I, P, P, P, P, P, F, P, F, P, P, P, I, I, I, I, F, U64, P, S],
  'stk_gn_bwd_out_ok': [I, I, I, I],
  'stk_conv2d_pl_ksplit': [I, I, I, I, I, I, I, I, I],
  'stk_conv2d_pl_halo': [I, I, I, I, I, I, I, I, I],
  'stk_gn_bwd_out_f32': [P, P, I, P, I, P, P, P, P, P, F, P, F, P, P, P, I, I, I, I, F, U64, P, P, F, P, F, P, I, P, S],
  'stk_gn_param_grad_batch': [P, I, I, S],
  'stk_conv2d_variant': [I, I, I, I, I, I, I, I, I, I, I, I, I, I],
  'stk_conv2d_fwd_ws_bytes': [I, I, I, I, I, I, I, I, I, I],
  'stk_conv2d_dgrad_ws_bytes': [I, I, I, I, I, I, I, I, I, I],
  'stk_conv2d_fwd_f32': [P, I, P, I, P, I, P, P, I, P, F, P, I, I, I, I, I, I, I, I, I, I, P, L, S],
  'stk_conv2d_dgrad_f32': [P, P, I, P, I, F, P, I, F, F, I, I, I, I, I, I, I, I, I, I, P, L, S],
  'stk_conv2d_wp_bytes': [I, I, I, I, I, I, I, I, I, I, I],
  'stk_conv2d_wp_desc': [I, P, I, I, I, I, I, P, P],
  'stk_conv2d_wprep_batch': [P, I, L, S],
  'stk_conv2d_fwd_wp_f32': [P, I, P, I, P, I, P, P, I, P, F, P, I, I, I, I, I, I, I, I, I, I, P, P, P, L, S],
  'stk_conv2d_fwd_rec_f32': [P, I, P, I, P, I, P, P, I, P, F, P, I, I, I, I, I, I, I, I, I, I, P, P, P, L, S],
  'stk_conv2d_dgrad_wp_f32': [P, P, I, P, I, F, P, I, F, F, I, I, I, I, I, I, I, I, I, I, P, P, P, L, S],
  'stk_conv2d_dgrad_rec_f32': [P, P, I, P, I, F, P, I, F, F, I, I, I, I, I, I, I, I, I, I, P, P, P, L, S],
  'stk_conv2d_wgrad_ws_bytes': [I, I, I, I, I, I, I, I],
  'stk_conv2d_wgrad_f32': [P, I, P, I, P, P, I, F, P, L, I, I, I, I, I, I, I, I, I, I, S],
  'stk_conv2d_wgrad_amax_f32': [P, I, P, I, P, P, I, F, P, L, I, I, I, I, I, I, I, I, I, I, P, I, S],
  'stk_gn_bound_f32': [P, P, I, I, I, F, P, S],
  'stk_gn_fwd_pl_f32': [P, I, P, I, P, P, P, P, P, P, P, I, I, I, F, I, F, U64, P, P, S],
  'stk_gn_fwd_pl_fused': [I, I, I, I],
  'stk_gn_fwd_pl_max_f32': [P, I, P, I, P, P, P, P, P, P, P, I, I, I, F, I, F, U64, P, P, P, P, S],
  'stk_planes_bytes': [I, I, I],
  'stk_amax_partial_f32': [P, L, P, S],
  'stk_split_planes_f32': [P, I, I, I, P, I, P, S],
  'stk_conv2d_pl_ok': [I, I, I, I, I, I, I, I, I, I, I],
  'stk_conv2d_fwd_pl_f32': [P, P, I, P, I, P, P, I, P, F, P, I, I, I, I, I, I, P, P, L, S],
  'stk_conv2d_dgrad_pl_f32': [P, P, P, I, P, I, F, P, I, F, F, I, I, I, I, I, I, P, P, L, S],
  'stk_conv2d_wgrad_pl_ok': [I, I, I, I, I],
  'stk_conv2d_wgrad_pl_ws_bytes': [I, I, I, I, I],
  'stk_conv2d_wgrad_pl_f32': [P, P, P, P, P, F, P, L, I, I, I, I, I, S],
  'stk_conv2d_wgrad_pl_wgs_f32': [P, P, P, P, P, F, P, L, I, I, I, I, I, I, S],
  'stk_bias_grad_f32': [P, I, I, I, F, P, I, P, P, S],
  'stk_bias_grad_amax_f32': [P, I, I, I, F, P, I, P, P, P, S],
  'stk_bias_grad_amax_res_f32': [P, I, I, I, F, P, I, P, P, P, F, P, S],
  'stk_bias_grad_amax_dual_f32': [P, I, I, I, F, P, I, P, P, P, P, P, S],
  'stk_gemm_f32': [P, L, L, L, P, L, L, L, P, L, L, L, P, I, I, I, I, I, F, F, S],
  'stk_softmax_fwd_f32': [P, P, L, I, F, S],
  'stk_softmax_bwd_f32': [P, P, P, L, I, F, S],
  'stk_attention_ok': [I, I, I],
  'stk_attention_fwd_f32': [P, P, P, L, P, P, P, I, I, I, F, S],
  'stk_attention_bwd_f32': [P, P, P, L, P, P, P, P, P, F, P, F, P, F, L, I, I, I, F, S],
  'stk_silu_fwd_f32': [P, P, L, S],
  'stk_silu_bwd_f32': [P, P, P, F, L, S],
  'stk_act_fwd_f32': [P, P, L, I, S],
  'stk_act_bwd_f32': [P, P, P, F, L, I, S],
  'stk_axpby_f32': [P, F, P, F, P, L, S],
  'stk_add_div_f32': [P, P, F, P, L, S],
  'stk_concat_f32': [P, I, P, I, P, I, I, S],
  'stk_concat_bwd_f32': [P, P, F, I, P, F, I, I, I, S],
  'stk_fixed_fourier_fwd_f32': [P, P, I, I, I, S],
  'stk_fixed_fourier_bwd_f32': [P, P, P, F, I, I, I, S],
  'stk_affine_f32': [P, F, F, P, L, S],
  'stk_fill_f32': [P, F, L, S],
  'stk_fill_strided_f32': [P, F, L, L, L, S],
  'stk_resample_naive_f32': [P, P, L, I, I, I, F, F, S],
  'stk_rowscale_f32': [P, P, P, I, L, I, S],
  'stk_timestep_embedding_f32': [P, P, P, I, I, S],
  'stk_fourier_embedding_f32': [P, P, P, I, I, S],
  'stk_perturb_f32': [P, P, P, P, P, I, L, S],
  'stk_sm_loss_fwd_f32': [P, P, P, P, P, I, L, I, I, I, S],
  'stk_sm_loss_bwd_f32': [P, P, P, P, P, P, I, L, I, I, I, S],
  'stk_sumsq_f32': [P, L, P, P, S],
  'stk_adam_f32': [P, P, P, P, L, F, F, F, F, F, I, F, F, P, F, S],
  'stk_adam_amsgrad_f32': [P, P, P, P, P, L, F, F, F, F, F, I, F, F, P, F, S],
  'stk_ema_f32': [P, P, L, F, S],
  'stk_dropout_mask_f32': [P, L, F, U64, S],
  'stk_samples_to_uint8': [P, P, I, I, L, S],
  'stk_preprocess_u8': [P, P, I, I, I, I, I, I, I, U64, S],
}
_RESTYPE = {'stk_strerror': c_char_p, 'stk_backend': c_char_p, 'stk_conv2d_wgrad_ws_bytes': c_long,
            'stk_conv2d_fwd_ws_bytes': c_long, 'stk_conv2d_dgrad_ws_bytes': c_long, 'stk_gn_ws_bytes': c_long,
            'stk_conv2d_wp_bytes': c_long, 'stk_conv2d_wp_desc': c_long, 'stk_planes_bytes': c_long, 'stk_conv2d_wgrad_pl_ws_bytes': c_long}
_NO_CHECK = set(_RESTYPE) | {'stk_version', 'stk_conv2d_variant', 'stk_conv2d_pl_ok', 'stk_gn_fwd_pl_fused', 'stk_conv2d_wgrad_pl_ok', 'stk_attention_ok', 'stk_gn_bwd_out_ok', 'stk_conv2d_pl_ksplit', 'stk_conv2d_pl_halo'}


class StkMissingError(RuntimeError):
  """The HIP library is not built / not loadable.  The product path never falls back."""


class StkError(RuntimeError):
  """A C-ABI entry returned a negative status."""


class StkLib:
  """A loaded implementation of include/stk.h with status-checked entry points."""

  def __init__(self, path):
    self.path = path
    self._cdll = ctypes.CDLL(path)
    for name, argtypes in SIGNATURES.items():
      try:
        fn = getattr(self._cdll, name)
      except AttributeError as e:
        raise StkMissingError(f'{path} does not export {name}') from e
      fn.argtypes = argtypes
      fn.restype = _RESTYPE.get(name, c_int)
      if name in _NO_CHECK:
        setattr(self, name[4:], fn)
      else:
        setattr(self, name[4:], self._checked(name, fn))
    self.backend = self._cdll.stk_backend().decode()
    self.is_device = self.backend.startswith('hip')

  def _checked(self, name, fn):
    strerror = self._cdll.stk_strerror

    def call(*args):
      rc = fn(*args)
      if rc != 0:
        raise StkError(f'{name} failed: {strerror(rc).decode()} (rc={rc})')
    call.__name__ = name
    call.raw = fn
    return call


_lock = threading.Lock()
_cache = {}


def load_path(path):
  path = os.path.abspath(path)
  with _lock:
    lib = _cache.get(path)
    if lib is None:
      if not os.path.exists(path):
        raise StkMissingError(f'{path} not found')
      lib = _cache[path] = StkLib(path)
    return lib


def load():
  """The product library.  Raises StkMissingError (never falls back) when it is absent.

  STK_LIBSTK=<path> (A/B runs of two BUILDS of the HIP library on one box, tools/insitu.sh) loads that file instead; it must be
  a HIP build -- a host / oracle library is refused, so the override cannot turn the product path into a CPU path."""
  alt = os.environ.get('STK_LIBSTK')
  if alt:
    lib = load_path(alt)
    if not lib.is_device:
      raise StkMissingError(f'STK_LIBSTK={alt} is not a HIP build of include/stk.h (backend {lib.backend}): refused')
    return lib
  if not os.path.exists(PRODUCT_LIB):
    raise StkMissingError(
      f'{PRODUCT_LIB} is not built. Run `python -c "import __graft_entry__ as g; g.build()"` '
      f'(or `make -C soft-truncation_amd/csrc`). The HIP library is mandatory: this package has '
      f'no CPU or PyTorch fallback for the score-network path.')
  return load_path(PRODUCT_LIB)


class _NoGuard:
  def __enter__(self):
    return self

  def __exit__(self, *exc):
    return False


_NO_GUARD = _NoGuard()


def device_guard(device):
  """Context that makes `device` the current HIP device for the raw C-ABI launches inside it.

  The kernels are launched through ctypes on torch's current stream of `device`; that handle is 0 (the NULL stream)
  unless a side stream is active, and the NULL stream belongs to whichever device is CURRENT -- so a model on
  cuda:1 in a process whose current device is cuda:0 would launch on the wrong GPU with another GPU's pointers.
  Every launch site of the package runs under this guard (no-op for host / oracle backends and when `device`
  already is current)."""
  import torch
  if device.type != 'cuda':
    return _NO_GUARD
  idx = device.index if device.index is not None else torch.cuda.current_device()
  if idx == torch.cuda.current_device():
    return _NO_GUARD
  return torch.cuda.device(idx)


def stream_ptr(device):
  """Raw hipStream_t of torch's current stream on `device` (0 for host / oracle backends)."""
  import torch
  if device.type != 'cuda':
    return 0
  return torch.cuda.current_stream(device).cuda_stream
