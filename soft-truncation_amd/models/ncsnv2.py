"""Placeholder for the reference's `models/ncsnv2.py` (imported by name at run_lib.py:23).

NCSN / NCSNv2 take `(config)` while `create_model` passes `(config, sde)` (models/ncsnv2.py:45 vs
models/utils.py:92), so they are unreachable in the reference and outside the hot path."""
from . import utils


def _unavailable(name):
  class _Model:
    def __init__(self, *args, **kwargs):
      raise NotImplementedError(f"model '{name}' is not constructible in the reference either (constructor arity); "
                                "every shipped config uses model.name = 'ncsnpp'")
  _Model.__name__ = name
  return utils.register_model(name=name)(_Model)


NCSNv2 = _unavailable('ncsnv2_64')
NCSN = _unavailable('ncsn')
NCSNv2_128 = _unavailable('ncsnv2_128')
NCSNv2_256 = _unavailable('ncsnv2_256')
