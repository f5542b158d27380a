"""soft-truncation_amd: the MI355X-native hot path of Kim-Dongjun/Soft-Truncation.

Score-network training step and SDE/ODE sampling loop behind the reference's own interfaces
(``models.utils`` registry, ``sde_lib.SDE``, ``losses.get_step_fn``, ``sampling.get_sampling_fn``,
``op.upfirdn2d``), executed by hand-written HIP kernels for gfx950 through the C ABI of
include/stk.h.  See DESIGN.md and INTEGRATION.md at the repository root.

The directory name contains a hyphen, so import it with
``importlib.import_module('soft-truncation_amd')`` or through the ``soft_truncation_amd`` alias
module at the repository root.
"""
import sys as _sys

from . import configs, datasets, sde_lib
from .engine import lib as _lib  # noqa: F401
from . import op
from .models import utils as _mutils  # noqa: F401
from .models import ema, layers, layerspp, ncsnpp, up_or_down_sampling  # noqa: F401
from . import models
from . import likelihood, losses, sampling, sampling_lib, utils

__all__ = ['configs', 'datasets', 'sde_lib', 'op', 'models', 'likelihood', 'losses', 'sampling', 'sampling_lib', 'utils',
           'install']

# name the reference's modules import under -> our module
_REFERENCE_NAMES = {
  'sde_lib': sde_lib,
  'losses': losses,
  'likelihood': likelihood,
  'sampling': sampling,
  'utils': utils,
  'op': op,
  'models': models,
}


def install():
  """Register this package's modules under the top-level names the reference's drivers import
  (``import sde_lib``, ``import losses``, ``from models import utils as mutils``, ``import op`` ...),
  so that an unmodified ``run_lib.py`` picks up the MI355X implementation.  See INTEGRATION.md."""
  pkg = __name__
  for short, mod in _REFERENCE_NAMES.items():
    _sys.modules[short] = mod
    prefix = mod.__name__ + '.'
    for full, sub in list(_sys.modules.items()):
      if full.startswith(prefix) and sub is not None:
        _sys.modules[short + '.' + full[len(prefix):]] = sub
  return pkg
