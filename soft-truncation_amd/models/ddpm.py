"""Placeholder for the reference's `models/ddpm.py`.

`run_lib.py:23` imports this module by name, so it has to exist for an unmodified driver.  The reference's
class cannot be constructed through `models.utils.create_model` (it calls `cls(config, sde)` while
`DDPM.__init__` takes `(config)`, models/ddpm.py:41 vs models/utils.py:92 -> TypeError) and no shipped config
names it, so it is outside the hot path; the registry entry exists and says so."""
from . import utils


@utils.register_model(name='ddpm')
class DDPM:
  def __init__(self, *args, **kwargs):
    raise NotImplementedError("model 'ddpm' is not constructible in the reference either (constructor arity); "
                              "every shipped config uses model.name = 'ncsnpp'")
