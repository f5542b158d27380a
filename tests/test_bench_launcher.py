"""`python bench.py --gpus N` must itself bring up N ranks when it is not already under a launcher (the round-1
bench parsed the flag and ran one process).  CPU check of that plumbing: --launch-check stops after the process
group is up (gloo here, RCCL on a GPU node) and the ranks are counted with one all-reduce."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(300)
def test_bench_gpus_2_launches_two_ranks():
  env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
  out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--launch-check'],
                       capture_output=True, text=True, env=env, timeout=280)
  assert out.returncode == 0, out.stderr[-2000:]
  lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
  assert len(lines) == 1, out.stdout
  rec = json.loads(lines[0])
  assert rec['n_gpus'] == 2 and rec['ranks_counted'] == 2 and rec['launch_check'] is True


def test_bench_single_process_stays_single():
  env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
  out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--launch-check'],
                       capture_output=True, text=True, env=env, timeout=120)
  assert out.returncode == 0, out.stderr[-2000:]
  rec = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][0])
  assert rec['n_gpus'] == 1 and rec['ranks_counted'] == 1
