"""Is the 256x256 / batch-4 training step host-bound?  Times step_fn with and without the trailing device sync, after an optional
earlier workload in the same process (the default bench.py run measures this net after the CIFAR-10 one).
  python tools/_probe/c5_host.py [--first cifar10] [--gc 0|1]"""
import argparse, gc, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
import soft_truncation_amd as st

ap = argparse.ArgumentParser()
ap.add_argument('--first', default='')
ap.add_argument('--gc', type=int, default=1)
ap.add_argument('--sampler', type=int, default=0)
ap.add_argument('--second', default='celebahq256')
args = ap.parse_args()
device = torch.device('cuda', 0)
torch.cuda.set_device(0)


def build(name):
  cfg_name, B, _ = bench.WORKLOADS[name]
  cfg = st.configs.get_config(cfg_name)
  cfg.device = device
  sde = st.sde_lib.get_sde(cfg, None)
  torch.manual_seed(0)
  model = st.models.utils.create_model(cfg, sde)
  model.module.engine().ensure_flat()
  opt = st.losses.get_optimizer(cfg, model.parameters())
  ema = st.models.ema.ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate)
  state = dict(optimizer=opt, model=model, ema=ema, step=0)
  fn = st.losses.get_step_fn(cfg, sde, train=True, optimize_fn=st.losses.optimization_manager(cfg))
  batch = st.datasets.synthetic_batch(cfg, B, device=device, generator=torch.Generator().manual_seed(4321))
  return cfg, sde, state, fn, batch, B


def measure(tag, state, fn, batch, B, n=12):
  for _ in range(6):
    fn(state, batch)
  torch.cuda.synchronize()
  for rep in range(3):
    host = []
    t0 = time.perf_counter()
    for _ in range(n):
      h0 = time.perf_counter()
      fn(state, batch)
      host.append(time.perf_counter() - h0)
    t_launch = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f'{tag} rep{rep}: {1e3 * t_all / n:.2f} ms/step; host loop alone {1e3 * t_launch / n:.2f} ms/step (min {1e3 * min(host):.2f} max {1e3 * max(host):.2f}); '
          f'gc counts {gc.get_count()} objects {len(gc.get_objects())}', flush=True)


if args.first:
  cfg, sde, state, fn, batch, B = build(args.first)
  measure(args.first, state, fn, batch, B, 10)
  if args.sampler:
    print(bench.sampler_rate(st, cfg, sde, state['model'], B, 6, device))
  del state, fn, batch
  torch.cuda.empty_cache()
if not args.gc:
  gc.collect(); gc.freeze(); gc.disable()
cfg, sde, state, fn, batch, B = build(args.second)
print('allocated GB', torch.cuda.memory_allocated() / 2**30, 'reserved GB', torch.cuda.memory_reserved() / 2**30)
measure(args.second, state, fn, batch, B)
print('allocated GB', torch.cuda.memory_allocated() / 2**30, 'reserved GB', torch.cuda.memory_reserved() / 2**30)
