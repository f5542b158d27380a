"""The watch test of the packed-fp32 hazard (tests/_model_cases.two_streams_deterministic) on a library built WITH the SLP
vectoriser (tools/_probe/build/libstk_slp.so, tools/_probe/build_variants.sh): it must FAIL -- a watch that passes on the
affected build watches nothing.   python tools/_probe/watch_with_slp.py [path of the library]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
  sys.path.insert(0, p)
from importlib import import_module
LIBMOD = import_module('soft-truncation_amd.engine.lib')
LIBMOD.PRODUCT_LIB = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'tools', '_probe', 'build', 'libstk_slp.so'))
import soft_truncation_amd as st
import _model_cases as cases
lib = st.engine.lib.load()
print('library:', lib.path, flush=True)
try:
  runs = cases.two_streams_deterministic(st, lib)
  print('PASSED', runs, 'backward passes bit-identical')
except AssertionError as e:
  print('FAILED (as an affected build must):', str(e)[:300])
