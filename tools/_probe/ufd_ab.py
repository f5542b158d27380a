"""upfirdn2d micro-benchmark on a chosen library build: PROBE_LIB=<so> python tools/_probe/ufd_ab.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
from importlib import import_module
LIBMOD = import_module('soft-truncation_amd.engine.lib')
if os.environ.get('PROBE_LIB'):
  LIBMOD.PRODUCT_LIB = os.path.abspath(os.environ['PROBE_LIB'])
sys.argv = ['bench_kernels.py', '--only', 'upfirdn', '--reps', '40']
import bench_kernels
bench_kernels.main()
