"""soft_truncation_amd.install(): the package's modules registered under the reference's top-level names, driven the
way run_lib.train drives them (tests/_install_driver.py, in a subprocess so the aliases do not leak into this
session).  CPU: on the checker backend; GPU (-m gpu): on the HIP library."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _drive(tmp_path, backend):
  out = subprocess.run([sys.executable, os.path.join(HERE, '_install_driver.py'), str(tmp_path), backend],
                       capture_output=True, text=True, timeout=600)
  assert out.returncode == 0, out.stderr[-3000:]
  rec = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
  assert rec['initial_step'] == 0 and rec['step'] == 3 and rec['resumed_step'] == 3
  assert rec['resumed_params_equal']
  assert rec['samples_shape'] == [4, 16, 16, 3] and rec['samples_dtype'] == 'uint8'
  assert all(m == m and m > 0 for m in rec['loss_means'])
  assert rec['optimizer'] == 'FusedAdam'
  return rec


@pytest.mark.timeout(700)
def test_install_drives_a_run_lib_shaped_loop_on_the_checker(tmp_path):
  assert _drive(tmp_path, 'checker')['backend'] == 'cpu-ref'


@pytest.mark.gpu
def test_install_drives_a_run_lib_shaped_loop_on_hip(tmp_path, hip_lib):      # hip_lib: skips without a GPU
  assert _drive(tmp_path, 'hip')['backend'] == 'hip-gfx950'
