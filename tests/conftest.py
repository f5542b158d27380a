import os
import subprocess
import sys

# every engine arena starts as NaN in the tests: an op that reads memory nobody wrote fails loudly (engine/executor.py)
os.environ.setdefault('STK_POISON', '1')

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
  if p not in sys.path:
    sys.path.insert(0, p)


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def st():
  import soft_truncation_amd
  return soft_truncation_amd


@pytest.fixture(scope='session')
def ref_lib(st):
  """The oracle's plain-C restatement (CPU checker).  Built on demand with gcc."""
  path = os.path.join(ROOT, 'oracle', 'libstk_ref.so')
  if not os.path.exists(path):
    subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle')])
  return st.engine.lib.load_path(path)


@pytest.fixture(scope='session')
def hip_lib(st):
  """The product library.  On a GPU box a missing library is a failure, never a skip."""
  import torch
  if os.environ.get('STK_SELFCHECK'):
    # harness self-check on a CPU-only machine: compare the checker with itself to debug the TEST code.  On a GPU box it
    # would turn every parity test into the oracle against itself, so there it is an error.
    if torch.cuda.is_available():
      raise RuntimeError('STK_SELFCHECK is a CPU-only harness check: unset it on a machine with a GPU')
    return st.engine.lib.load_path(os.path.join(ROOT, 'oracle', 'libstk_ref.so'))
  if not torch.cuda.is_available():
    pytest.skip('no GPU in this container')
  return st.engine.lib.load()
