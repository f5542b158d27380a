"""Subprocess body of the install() tests: what an unmodified run_lib.train does with the modules it imports
(reference run_lib.py:28-31, 51-110), executed through the reference's TOP-LEVEL module names after
``soft_truncation_amd.install()`` has registered this package under them.

usage: python _install_driver.py <workdir> <hip|checker>
Prints one JSON line.  `checker` (CPU self-test of this script and of install()) routes engine.lib.load() to the
oracle's C library; `hip` is the product."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
  if p not in sys.path:
    sys.path.insert(0, p)

import numpy as np
import torch


def main(workdir, backend):
  import soft_truncation_amd as st
  if backend == 'checker':
    checker = st.engine.lib.load_path(os.path.join(ROOT, 'oracle', 'libstk_ref.so'))
    st.engine.lib.load = lambda: checker
    device = torch.device('cpu')
  else:
    device = torch.device('cuda:0')
  st.install()

  # ---- from here on: the reference driver's imports and call sequence, top-level names only -------------------
  import sde_lib
  import losses                                    # noqa: F401  (run_lib.py:31)
  import sampling                                  # noqa: F401
  import utils
  import likelihood                                # noqa: F401
  from models import utils as mutils
  from models import ncsnpp                        # noqa: F401  (registers the model, run_lib.py:24)
  from models import ema as ema_mod                # noqa: F401
  import op
  assert sde_lib is st.sde_lib and utils is st.utils and mutils is st.models.utils and op is st.op
  assert 'ncsnpp' in mutils._MODELS

  config = st.configs.tiny(st.configs.cifar10_ddpmpp_nll_st())
  config.device = device
  config.model.num_scales = 4
  config.optim.warmup = 2
  config.sampling.method, config.sampling.predictor, config.sampling.corrector = 'pc', 'euler_maruyama', 'none'
  config.sampling.batch_size = 4
  config.training.batch_size = 4
  sample_dir = os.path.join(workdir, 'samples')
  os.makedirs(sample_dir, exist_ok=True)

  torch.manual_seed(0)
  np.random.seed(0)
  sde = sde_lib.get_sde(config, None)
  state, score_model, ema, checkpoint_dir, checkpoint_meta_dir = utils.load_model(config, workdir, sde=sde)
  initial_step = int(state['step'])
  scaler = st.datasets.get_data_scaler(config)
  inverse_scaler = st.datasets.get_data_inverse_scaler(config)
  train_step_fn, nll_fn, nelbo_fn, sampling_fn = utils.get_loss_fns(config, sde, inverse_scaler)
  gen = torch.Generator().manual_seed(1)
  means = []
  for step in range(initial_step, initial_step + 3):
    batch = torch.rand(4, 3, config.data.image_size, config.data.image_size, generator=gen).to(device)
    if config.data.dequantization == 'uniform':
      batch = (255. * batch + torch.rand_like(batch)) / 256.
    batch = scaler(batch)
    losses_ = train_step_fn(state, batch)
    assert losses_.device.type == 'cpu' and losses_.shape == (4,)
    means.append(float(torch.mean(losses_)))
  utils.save_checkpoint(config, checkpoint_meta_dir, state)
  ema.store(score_model.parameters())
  ema.copy_to(score_model.parameters())
  samples = st.sampling_lib.get_samples(config, score_model, state, sampling_fn, state['step'], 7, sample_dir)
  ema.restore(score_model.parameters())
  # a restarted driver resumes from the meta checkpoint
  state2, _, _, _, _ = utils.load_model(config, workdir, sde=sde)
  same = all(torch.equal(a.detach().cpu(), b.detach().cpu())
             for a, b in zip(state['model'].parameters(), state2['model'].parameters()))
  print(json.dumps({'initial_step': initial_step, 'step': int(state['step']), 'resumed_step': int(state2['step']),
                    'resumed_params_equal': bool(same), 'loss_means': means,
                    'samples_shape': list(samples.shape), 'samples_dtype': str(samples.dtype),
                    'backend': st.engine.lib.load().backend,
                    'optimizer': type(state['optimizer']).__name__}), flush=True)


if __name__ == '__main__':
  main(sys.argv[1], sys.argv[2])
