"""Does the 256x256 / batch-4 step time depend on where the arenas land?  Rebuilds the model with a growing dummy allocation in
front of it and prints step time + arena base addresses.   python tools/_probe/c5_addr.py [--workload celebahq256]"""
import argparse, gc, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
import soft_truncation_amd as st

ap = argparse.ArgumentParser()
ap.add_argument('--workload', default='celebahq256')
ap.add_argument('--pads', default='0,2,4,6,8,10,14,18,22,26,30,34,66,130')
args = ap.parse_args()
device = torch.device('cuda', 0)
torch.cuda.set_device(0)
cfg_name, B, _ = bench.WORKLOADS[args.workload]
for pad_mb in [int(v) for v in args.pads.split(',')]:
  pads = [torch.empty(2 << 20, dtype=torch.uint8, device=device) for _ in range(pad_mb // 2)]   # separate 2 MB blocks
  cfg = st.configs.get_config(cfg_name)
  cfg.device = device
  sde = st.sde_lib.get_sde(cfg, None)
  torch.manual_seed(0)
  model = st.models.utils.create_model(cfg, sde)
  eng = model.module.engine()
  eng.ensure_flat()
  opt = st.losses.get_optimizer(cfg, model.parameters())
  ema = st.models.ema.ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate)
  state = dict(optimizer=opt, model=model, ema=ema, step=0)
  fn = st.losses.get_step_fn(cfg, sde, train=True, optimize_fn=st.losses.optimization_manager(cfg))
  batch = st.datasets.synthetic_batch(cfg, B, device=device, generator=torch.Generator().manual_seed(4321))
  for _ in range(6):
    fn(state, batch)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(10):
    fn(state, batch)
  torch.cuda.synchronize()
  ms = 1e2 * (time.perf_counter() - t0)
  prog = next(iter(eng.programs.values()))
  ctxs = list(prog.free) + [c for c in eng._awaiting]
  c = ctxs[0] if ctxs else None
  def a(t):
    return 'None' if t is None else hex(t.data_ptr())
  print(f'pad {pad_mb:4d} MB: {ms:6.2f} ms/step  flat {a(eng.flat.data)} grad {a(eng.flat.grad)} ws {a(prog.ws)} ws2 {a(prog.ws2)} '
        + (f'act {a(c.act)} gact {a(c.gact)} pl {a(c.pl)}' if c is not None else 'no ctx'), flush=True)
  del state, fn, batch, model, eng, opt, ema, prog, ctxs, c
  gc.collect()
  torch.cuda.empty_cache()
  del pads
  torch.cuda.empty_cache()
