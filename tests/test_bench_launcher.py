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


@pytest.mark.timeout(600)
def test_bench_main_two_ranks_emits_one_short_line(tmp_path):
  """bench.py's real main() at WORLD_SIZE = 2 through to the emitted stdout line, on the CPU checker over gloo
  (tests/_bench_worker.py replaces four seams only): n_gpus = 2, value = GLOBAL images/s, the line is short and alone."""
  import socket
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'oracle')])
  base = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
  base.update(WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), BENCH_DETAIL=str(tmp_path / 'detail.json'),
              STK_POISON='1')
  procs = []
  for rank in range(2):
    env = dict(base, RANK=str(rank), LOCAL_RANK=str(rank))
    procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', '_bench_worker.py')], env=env,
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
  outs = [p.communicate(timeout=560) for p in procs]
  for p, (so, se) in zip(procs, outs):
    assert p.returncode == 0, se[-3000:]
  assert outs[1][0] == '', outs[1][0]                       # only rank 0 writes to stdout
  stdout0 = outs[0][0]
  assert stdout0.count('\n') == 1 and len(stdout0) < 4000, stdout0
  line = json.loads(stdout0)
  assert line['n_gpus'] == 2 and line['config']['parallelism'] == 'dp2'
  assert line['config']['per_gpu_batch'] == 4 and line['config']['global_batch'] == 8
  assert line['steps'] == 3 and line['warmup'] == 1 and line['scaling'] == 'weak' and line['unit'] == 'images/s'
  # value is the whole job's rate: global batch x steps / max-over-ranks elapsed
  assert line['value'] == pytest.approx(8 * 1e3 / line['ms_per_step'], rel=1e-3)
  # no GPU here: the exchange-stream probe cannot run, and the line says so instead of staying silent
  assert line['exchange_serialised'] is True and 'error' in line['exchange_stream']
  detail = json.load(open(line['detail']))
  assert detail['n_gpus'] == 2 and detail['value'] == pytest.approx(line['value'], rel=1e-3)
  assert len(outs[0][1]) < 4000, outs[0][1][-4000:]         # stderr stays short too (no progress bars, no tables)
