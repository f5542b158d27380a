"""One rank of `bench.py`'s REAL main() at WORLD_SIZE = 2 without a GPU (tests/test_bench_launcher.py): gloo process group, CPU
tensors, the oracle's C restatement injected as the engine's library, a fixture-size net in place of the 61.8 M-parameter one.
Only bench.py's four seams are replaced (device_of, init_group, device_sync, build_training) -- argument parsing, seeding, the
sharded batch, priming / warm-up / timed steps between barriers, the MAX over ranks, the global images/s and the emitted stdout
line are bench.py's own code.  RANK / WORLD_SIZE / MASTER_* come from the environment, as under torch.distributed.run."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
  if p not in sys.path:
    sys.path.insert(0, p)

import torch
import torch.distributed as dist


def main():
  torch.set_num_threads(2)
  import importlib.util
  spec = importlib.util.spec_from_file_location('bench_main', os.path.join(ROOT, 'bench.py'))
  bench = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(bench)
  import soft_truncation_amd as st
  from _model_util import randomize_, tiny_config
  lib = st.engine.lib.load_path(os.path.join(ROOT, 'oracle', 'libstk_ref.so'))

  real_get = st.configs.get_config
  st.configs.get_config = lambda name: tiny_config(st, 'vp', device='cpu') if name == 'tiny_vp' else real_get(name)
  bench.WORKLOADS['cifar10'] = ('tiny_vp', 4, 'fixture-size DDPM++ (VP) on the CPU checker (test harness)')
  bench.TRAIN_FLOPS_PER_IMG['tiny_vp'] = 1e9

  def build_training(st_, cfg, sde):
    cfg.optim.warmup = 2
    net = st.models.ncsnpp.NCSNpp(cfg, sde)
    net.set_backend(lib)
    net = net.to(cfg.device)
    randomize_(net, 0)
    model = st.models.utils.DataParallel(net)
    net.engine().ensure_flat()
    opt = st.losses.get_optimizer(cfg, model.parameters())
    opt._backend = lib
    ema = st.models.ema.ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate)
    ema.set_backend(lib)
    state = dict(optimizer=opt, model=model, ema=ema, step=0)
    return state, st.losses.get_step_fn(cfg, sde, train=True, optimize_fn=st.losses.optimization_manager(cfg))

  bench.device_of = lambda local_rank: torch.device('cpu')
  bench.init_group = lambda device: dist.init_process_group('gloo')
  bench.device_sync = lambda: None
  bench.build_training = build_training
  calls = []
  real_all_reduce = dist.all_reduce
  dist.all_reduce = lambda *a, **k: (calls.append(a[0].numel()), real_all_reduce(*a, **k))[1]
  sys.argv = ['bench.py', '--gpus', os.environ['WORLD_SIZE'], '--steps', '3', '--warmup', '1', '--no-kernel-timer',
              '--detail', os.environ['BENCH_DETAIL']]
  bench.main()
  # the gradient exchange really ran in every step: more all-reduces than the one MAX over the elapsed times
  print(f'rank {os.environ["RANK"]}: {len(calls)} all-reduces', file=sys.stderr)
  assert len(calls) >= 1 + (2 + 1 + 3), calls


if __name__ == '__main__':
  main()
