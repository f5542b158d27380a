"""The stdout line of bench.py must fit the driver: round 5's 20 KB line overflowed the 8000 characters of stdout the driver keeps
and arrived headless (BENCH_r05.json: parsed = null).  short_line() is checked here on a canned record of everything a run measures
(tests/golden/bench_out_r05.json = the round-5 run's full object: data, per-kernel tables of three workloads included) for N = 1 and
for an N = 8 record, and on worst-case strings."""
import copy
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def bench():
  import importlib.util
  spec = importlib.util.spec_from_file_location('bench_under_test', os.path.join(ROOT, 'bench.py'))
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  return mod


@pytest.fixture()
def canned():
  out = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'bench_out_r05.json')))
  out['roofline']['shares_chip'] = False
  out['roofline_best'] = copy.deepcopy(out['roofline'])
  return out


CONTRACT = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
            'dtype', 'data', 'config')


def test_single_gpu_line_is_short_and_complete(bench, canned):
  text = bench.short_line(canned, 'cifar10', 'gpurun_out/bench_detail.json')
  assert len(text) < bench.LINE_LIMIT == 4000, len(text)
  assert '\n' not in text
  line = json.loads(text)
  for k in CONTRACT:
    assert k in line, k
  assert line['config']['workload'].startswith('DDPM++ (VP) CIFAR-10') and 'model' not in line['config']
  roof = line['roofline']
  for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel', 'avg_us', 'shares_chip'):
    assert k in roof, k
  assert abs(roof['frac'] - roof['achieved'] / roof['peak']) < 1e-3
  cb = line['cpu_baseline']
  for k in ('value', 'unit', 'cores', 'kind', 'sample', 'sec_per_step', 'batch128'):
    assert k in cb, k
  assert set(line['headline']) == {'cifar10', 'celeba64', 'celebahq256'}
  assert line['headline']['celebahq256']['sampler_N1000']['full_run_s'] > 0
  assert line['detail'] == 'gpurun_out/bench_detail.json'
  assert line['value'] == pytest.approx(canned['value'], rel=1e-6)
  # the tables live in the side file only
  assert 'kernels' not in line and 'workloads' not in line and 'workload_kernels' not in line


def test_eight_gpu_line_is_short_and_flags_a_serialised_exchange(bench, canned):
  out = {k: v for k, v in canned.items() if k not in ('workloads', 'cpu_baseline', 'parity_probe', 'sampler', 'exchange_proxy',
                                                     'arithmetic_check')}
  out.update(n_gpus=8, value=8 * canned['value'])
  out['config'].update(global_batch=1024, parallelism='dp8')
  out['exchange_stream'] = {'beside_main': True, 'beside_side': False, 'ok': False, 'collective_ms': [0.06, 39.7], 'spin_ms': [39.8, 39.8],
                            'attempts': 1, 'pool_steered': True, 'ranks': 8, 'note': 'x' * 500}
  text = bench.short_line(out, 'cifar10', None)
  assert len(text) < bench.LINE_LIMIT
  line = json.loads(text)
  assert line['n_gpus'] == 8 and line['config']['parallelism'] == 'dp8' and line['roofline']['frac'] > 0
  assert line['exchange_stream']['ok'] is False and line['exchange_serialised'] is True
  out['exchange_stream']['ok'] = True
  assert 'exchange_serialised' not in json.loads(bench.short_line(out, 'cifar10', None))


def test_line_stays_short_with_errors_and_long_strings(bench, canned):
  out = copy.deepcopy(canned)
  out['workloads']['celeba64'] = {'error': 'E' * 5000}
  out['parity_probe'] = {'error': 'P' * 5000}
  out['cpu_baseline']['sample'] = 'S' * 5000
  out['exchange_stream'] = {'error': 'X' * 5000}
  text = bench.short_line(out, 'cifar10', 'd.json')
  assert len(text) < bench.LINE_LIMIT
  assert json.loads(text)['headline']['celeba64']['error'].startswith('E')


def test_roofline_is_the_kernel_with_the_largest_total_time(bench):
  summ = {'conv3x3.fwd.x2p.h32': {'total_ms': 11.2, 'tflops': 330.0, 'avg_us': 187.0, 'count': 60, 'flops_per_launch': 61.8e9},
          'conv3x3.wgrad.x2p.w16': {'total_ms': 14.0, 'tflops': 203.0, 'avg_us': 212.0, 'count': 66, 'flops_per_launch': 43.0e9},
          'conv1x1.fwd.t64': {'total_ms': 0.3, 'tflops': 150.0, 'avg_us': 20.0, 'count': 15, 'flops_per_launch': 1e9}}
  roof, best, kernels = bench.roofline_of(summ, 3, 36.5, workload='no-such-workload')
  assert roof['kernel'] == 'conv3x3.wgrad.x2p.w16' and roof['shares_chip'] is True and roof['traffic'] is None
  assert roof['frac'] == pytest.approx(203.0 / (2500.0 / 3))
  # the tiny f32-input kernel has the highest fraction of ITS peak but takes < 2 % of the step
  assert best['kernel'] == 'conv3x3.fwd.x2p.h32' and best['shares_chip'] is False
  assert kernels['conv3x3.wgrad.x2p.w16']['shares_chip'] is True


def test_detail_file_round_trip(bench, canned, tmp_path):
  p = bench.write_detail(canned, str(tmp_path / 'sub' / 'detail.json'))
  assert p and json.load(open(p))['kernels'] == canned['kernels']
