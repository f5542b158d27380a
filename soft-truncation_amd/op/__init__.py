"""Native-op surface of the reference (`op/__init__.py:1-2`) on the MI355X C ABI."""
from .fused_act import FusedLeakyReLU, fused_leaky_relu
from .upfirdn2d import upfirdn2d
