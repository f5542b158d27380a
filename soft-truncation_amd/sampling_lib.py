"""Sample post-processing and on-disk format -- the step right after the sampling loop.

Mirror of the reference's ``sampling_lib.get_dir_name`` / ``get_samples`` (sampling_lib.py:28-57): uint8 NHWC samples
in ``samples_{r}.npz`` (key ``samples``, compressed) and a ``sample_{r}.png`` grid with torchvision's
``make_grid(nrow=int(sqrt(n)), padding=2)`` / ``save_image`` layout.  The float -> uint8 NHWC conversion runs on the
device (``stk_samples_to_uint8``) so a quarter of the bytes cross PCIe; torchvision and TensorFlow's gfile are not
needed (plain files; the PNG is written with Pillow when it is installed).  The Inception / FID half of the
reference's file (get_latents, compute_fid_and_is, ...) is evaluation, outside the hot path.
"""
import io
import logging
import os

import numpy as np
import torch

from .engine import lib as stk_lib


def get_dir_name(config, sample_dir, step):
  """sampling_lib.py:28-33."""
  s = config.sampling
  if s.method == 'pc':
    return os.path.join(sample_dir, f'iter_{step}_{s.truncation_time}_{s.noise_removal}_{s.predictor}_{s.corrector}_{s.snr}')
  return os.path.join(sample_dir, f'iter_{step}_{s.truncation_time}_{s.noise_removal}')


def samples_to_uint8(samples, backend=None):
  """``np.clip(samples.permute(0, 2, 3, 1).cpu().numpy() * 255., 0, 255).astype(np.uint8)`` (sampling_lib.py:43).

  ``samples``: float32 ``[N, C, H, W]`` in [0, 1] (after the inverse scaler).  Device tensors are converted by the
  HIP kernel; ``backend`` lets a test run the same call on the oracle's checker library."""
  x = samples.detach().to(torch.float32).contiguous()
  N, C, H, W = x.shape
  lib = backend if backend is not None else stk_lib.load()
  if lib.is_device != (x.device.type == 'cuda'):
    raise RuntimeError(f'backend {lib.backend} cannot convert samples on {x.device}')
  out = torch.empty((N, H, W, C), dtype=torch.uint8, device=x.device)
  with stk_lib.device_guard(x.device):
    lib.samples_to_uint8(x.data_ptr(), out.data_ptr(), N, C, H * W, stk_lib.stream_ptr(x.device))
  return out.cpu().numpy()


def make_grid_uint8(samples, nrow, padding=2):
  """torchvision.utils.make_grid(images / 255., nrow, padding) followed by save_image's uint8 conversion, for uint8
  NHWC input: images left to right, top to bottom, `padding` black pixels around every image."""
  n, h, w, c = samples.shape
  xmaps = min(nrow, n)
  ymaps = int(np.ceil(n / xmaps))
  gh, gw = h + padding, w + padding
  grid = np.zeros((gh * ymaps + padding, gw * xmaps + padding, c), dtype=np.uint8)
  for k in range(n):
    y, x = divmod(k, xmaps)
    grid[y * gh + padding:y * gh + padding + h, x * gw + padding:x * gw + padding + w] = samples[k]
  if c == 1:          # make_grid repeats single-channel images to RGB
    grid = np.repeat(grid, 3, axis=2)
  return grid


def get_samples(config, score_model, state, sampling_fn, step, r, sample_dir):
  """sampling_lib.py:36-57: sample (unless ``samples_{r}.npz`` already exists), post-process, write npz + grid."""
  logging.info('sampling -- ckpt step: %d, round: %d' % (step, r))
  dir_name = get_dir_name(config, sample_dir, step)
  os.makedirs(dir_name, exist_ok=True)
  path = os.path.join(dir_name, f'samples_{r}.npz')
  if os.path.exists(path):
    return np.load(path)['samples']
  samples, _ = sampling_fn(score_model)
  samples = samples_to_uint8(samples)
  samples = samples.reshape((-1, config.data.image_size, config.data.image_size, config.data.num_channels))
  with open(path, 'wb') as fout:
    io_buffer = io.BytesIO()
    np.savez_compressed(io_buffer, samples=samples)
    fout.write(io_buffer.getvalue())
  grid = make_grid_uint8(samples, int(np.sqrt(samples.shape[0])), padding=2)
  try:
    from PIL import Image
    Image.fromarray(grid).save(os.path.join(dir_name, f'sample_{r}.png'), format='png')
  except ImportError:            # the grid is a convenience image; the npz is the product
    logging.warning('Pillow is not installed: sample_%d.png not written' % r)
  return samples
