"""GroupNorm backward with non-temporal stores / loads (variant libraries built by hand into tools/_probe/build): PROBE_LIB=<so>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
from importlib import import_module
LIBMOD = import_module('soft-truncation_amd.engine.lib')
if os.environ.get('PROBE_LIB'):
  LIBMOD.PRODUCT_LIB = os.path.abspath(os.environ['PROBE_LIB'])
sys.argv = ['bench_kernels.py', '--only', 'gn', '--reps', '30']
import bench_kernels
bench_kernels.main()
