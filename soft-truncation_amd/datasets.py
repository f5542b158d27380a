"""Data scalers of the reference (datasets.py:56-71) and the device-side tail of its input pipeline.

The tf.data / tfds readers themselves are outside the hot path; what they hand over per step -- uint8 images -- is
turned into the float batch ``step_fn`` takes by one kernel (`device_batch`): uint8 -> [0, 1], random left-right flip,
uniform dequantisation and the scaler, i.e. datasets.py:313-324 plus run_lib.py:72-75, with a quarter of the bytes
crossing PCIe.  Benchmarks and tests feed synthetic batches of the configured shape."""
import torch

from .engine import lib as stk_lib


def get_data_scaler(config):
  """[0,1] -> [-1,1] when ``data.centered`` else identity (datasets.py:56-62)."""
  if config.data.centered:
    return lambda x: x * 2. - 1.
  return lambda x: x


def get_data_inverse_scaler(config):
  """Inverse of ``get_data_scaler`` (datasets.py:65-71)."""
  if config.data.centered:
    return lambda x: (x + 1.) / 2.
  return lambda x: x


def synthetic_batch(config, batch_size, device=None, generator=None):
  """x ~ U[0,1) per pixel then the config's scaler (SURVEY.md 8(d) synthetic inputs)."""
  d = config.data
  x = torch.rand(batch_size, d.num_channels, d.image_size, d.image_size, generator=generator)
  if device is not None:
    x = x.to(device)
  return get_data_scaler(config)(x)


def device_batch(config, images_u8, seed, evaluation=False, backend=None):
  """uint8 ``[N, H, W, C]`` images (device tensor) -> float32 ``[N, C, H, W]`` batch for ``step_fn``.

  Equivalent to the reference's ``convert_image_dtype`` + ``random_flip_left_right`` (training only, when
  ``data.random_flip``) + ``(255 x + rand) / 256`` (when ``data.dequantization == 'uniform'``) + scaler.  The random
  draws come from the library's counter RNG (include/stk_rng.h), a pure function of ``seed`` -- pass a fresh seed
  per step.  ``backend`` lets a test run the same call on the oracle's checker library."""
  d = config.data
  x = images_u8.contiguous()
  if x.dtype != torch.uint8 or x.dim() != 4:
    raise ValueError('images_u8 must be a uint8 [N, H, W, C] tensor')
  N, H, W, C = x.shape
  lib = backend if backend is not None else stk_lib.load()
  if lib.is_device != (x.device.type == 'cuda'):
    raise RuntimeError(f'backend {lib.backend} cannot preprocess a batch on {x.device}')
  out = torch.empty((N, C, H, W), dtype=torch.float32, device=x.device)
  flip = int(bool(d.random_flip) and not evaluation)
  dequant = int(getattr(d, 'dequantization', 'none') == 'uniform')
  with stk_lib.device_guard(x.device):
    lib.preprocess_u8(x.data_ptr(), out.data_ptr(), N, C, H, W, flip, dequant, int(bool(d.centered)),
                      int(seed) & 0xFFFFFFFFFFFFFFFF, stk_lib.stream_ptr(x.device))
  return out
