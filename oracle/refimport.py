"""ORACLE (test infrastructure): import the reference's own Python CPU path in THIS container.

/root/reference is a PyTorch research repo whose hot path imports cleanly on CPU once a few
absent third-party modules are stubbed.  This module performs that
stubbing and returns the reference's modules so that

* tools/make_golden.py can generate the fixtures under tests/golden/ (inputs + the reference's
  outputs), and
* tests/test_ref_live.py can compare the restatements with the live reference when it is present.

Nothing here travels anywhere: /root/reference does not exist on the GPU box, so every user of
this module must be skipped when ``available()`` is False.  No reference source is copied.
"""
import importlib
import importlib.machinery
import os
import sys
import types

REFERENCE_ROOT = '/root/reference'


def available():
  return os.path.isdir(os.path.join(REFERENCE_ROOT, 'models'))


class _AttrDict(dict):
  """Minimal ml_collections.ConfigDict stand-in (attribute access + to_dict)."""

  def __getattr__(self, k):
    try:
      return self[k]
    except KeyError as e:
      raise AttributeError(k) from e

  def __setattr__(self, k, v):
    self[k] = v

  def to_dict(self):
    return dict(self)


def _stub(name, **attrs):
  m = types.ModuleType(name)
  m.__spec__ = importlib.machinery.ModuleSpec(name, None)
  m.__dict__.update(attrs)
  sys.modules[name] = m
  return m


_loaded = None


def load():
  """Returns a namespace with the reference's sde_lib, losses, sampling, models.*, op, configs."""
  global _loaded
  if _loaded is not None:
    return _loaded
  if not available():
    raise RuntimeError('/root/reference is not present (it never is on the GPU box)')
  import torch
  import torch.utils.cpp_extension as cpp_ext

  saved = {k: sys.modules.get(k) for k in
           ('sde_lib', 'losses', 'sampling', 'likelihood', 'models', 'op', 'configs', 'utils')}
  for k in list(sys.modules):
    if k.split('.')[0] in saved:
      del sys.modules[k]

  _stub('ml_collections', ConfigDict=_AttrDict)
  gfile = types.SimpleNamespace(exists=os.path.exists, makedirs=lambda p: os.makedirs(p, exist_ok=True))
  _stub('tensorflow', io=types.SimpleNamespace(gfile=gfile), float32='float32', function=lambda f: f)
  for n in ('tensorflow_datasets', 'tensorflow_gan', 'tensorflow_hub', 'natsort', 'six',
            'torchvision', 'torchvision.utils', 'torchvision.transforms', 'absl'):
    if n not in sys.modules:
      _stub(n)
  sys.modules['torchvision.utils'].make_grid = lambda *a, **k: None
  sys.modules['torchvision.utils'].save_image = lambda *a, **k: None
  if 'absl.flags' not in sys.modules:
    _stub('absl.flags', FLAGS=types.SimpleNamespace())

  real_load = cpp_ext.load
  cpp_ext.load = lambda *a, **k: None          # op/*.py JIT-build CUDA at import time
  sys.path.insert(0, REFERENCE_ROOT)
  try:
    ns = types.SimpleNamespace()
    ns.sde_lib = importlib.import_module('sde_lib')
    ns.op = importlib.import_module('op')
    ns.op_upfirdn2d = importlib.import_module('op.upfirdn2d')
    ns.op_fused_act = importlib.import_module('op.fused_act')
    ns.mutils = importlib.import_module('models.utils')
    ns.ncsnpp = importlib.import_module('models.ncsnpp')
    ns.layers = importlib.import_module('models.layers')
    ns.layerspp = importlib.import_module('models.layerspp')
    ns.uds = importlib.import_module('models.up_or_down_sampling')
    ns.ema = importlib.import_module('models.ema')
    ns.losses = importlib.import_module('losses')
    ns.sampling = importlib.import_module('sampling')
    ns.likelihood = importlib.import_module('likelihood')
    ns.config_module = lambda dotted: importlib.import_module(dotted)
    ns.modules = {k: v for k, v in sys.modules.items()
                  if k.split('.')[0] in saved and v is not None}
  finally:
    cpp_ext.load = real_load
    sys.path.remove(REFERENCE_ROOT)
    # leave the reference's modules registered only inside the namespace object: unregister the
    # top-level names again so they cannot shadow this repository's modules of the same name
    for k in list(sys.modules):
      if k.split('.')[0] in saved:
        del sys.modules[k]
    for k, v in saved.items():
      if v is not None:
        sys.modules[k] = v
  _loaded = ns
  return ns


def get_config(dotted, device='cpu'):
  """e.g. get_config('configs.vp.CIFAR10.ddpmpp_nll_st')"""
  import torch
  ns = load()
  # config modules import `configs.default_*` absolutely: needs the reference root on sys.path
  sys.path.insert(0, REFERENCE_ROOT)
  stash = {k: sys.modules.pop(k) for k in list(sys.modules) if k.split('.')[0] == 'configs'}
  try:
    cfg = importlib.import_module(dotted).get_config()
  finally:
    for k in list(sys.modules):
      if k.split('.')[0] == 'configs':
        del sys.modules[k]
    sys.modules.update(stash)
    sys.path.remove(REFERENCE_ROOT)
  cfg.device = torch.device(device)
  return cfg
