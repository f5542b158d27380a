"""Data scalers of the reference (datasets.py:56-71).  The input pipelines themselves
(tf.data / tfds / torchvision) are outside the hot path; benchmarks and tests feed synthetic
batches of the configured shape."""
import torch


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
