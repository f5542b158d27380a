"""Score-network model family (reference: models/).  Importing the package registers 'ncsnpp'."""
from . import utils, ema, layers, layerspp, up_or_down_sampling, ncsnpp, ddpm, ncsnv2  # noqa: F401
