"""Importable alias for the ``soft-truncation_amd`` package (a hyphen is not a valid identifier).

``import soft_truncation_amd as st`` gives the real package object; every already-imported
submodule is aliased too, so ``from soft_truncation_amd.models import utils`` returns the same
module objects (one model registry, one backend binding)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
  sys.path.insert(0, _root)
_pkg = importlib.import_module('soft-truncation_amd')
for _name, _mod in list(sys.modules.items()):
  if _name.startswith('soft-truncation_amd.') and _mod is not None:
    sys.modules['soft_truncation_amd' + _name[len('soft-truncation_amd'):]] = _mod
sys.modules[__name__] = _pkg
