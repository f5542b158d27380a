"""GroupNorm kernel probe (diagnostic, runs on the GPU box): forward (fp32 output and planes output), backward, with and
without SiLU / dropout, on the shapes of the DDPM++ 32x32 step; GB/s are algorithmic bytes (fwd: read x + write y or
planes; bwd: read x, dy + write dx) over the measured time; a device copy of the same tensor is timed beside them."""
import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from _util import call
from importlib import import_module
lib = import_module('soft-truncation_amd.engine.lib').load()
d = torch.device('cuda')
def timeit(f, n=30):
  for _ in range(5): f()
  torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
  s.record()
  for _ in range(n): f()
  e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
N, G = 128, 32
only = sys.argv[1] if len(sys.argv) > 1 else ''
for C1, C2, H in ((128, 0, 32), (256, 0, 16), (256, 0, 8), (256, 0, 4), (128, 128, 32), (256, 256, 16)):
  C = C1 + C2
  x1 = torch.randn(N, C1, H, H, device=d); x2 = torch.randn(N, C2, H, H, device=d) if C2 else None
  g = torch.ones(C, device=d); b = torch.zeros(C, device=d)
  y = torch.empty(N, C, H, H, device=d); mean = torch.empty(N * G, device=d); rstd = torch.empty(N * G, device=d)
  dy = torch.randn_like(y); dx1 = torch.empty_like(x1); dx2 = torch.empty_like(x2) if C2 else None
  dg = torch.zeros(C, device=d); db = torch.zeros(C, device=d)
  ws = torch.empty(max(int(lib.gn_ws_bytes(N, C, H * H, G)), 8 * N * C) // 4 + 64, device=d)
  planes = torch.empty(int(lib.planes_bytes(N, C, H * H)), dtype=torch.uint8, device=d); rec = torch.empty(256, device=d)
  nb = y.numel() * 4
  tc = timeit(lambda: y.copy_(dy))
  print(f'C{C1}+{C2}@{H}: copy {tc:6.1f} us ({2*nb/tc/1e3:5.0f} GB/s)')
  for act, p in ((1, 0.1), (1, 0.0), (0, 0.0)):
    tf = timeit(lambda: call(lib, 'gn_fwd_f32', x1, C1, x2, C2, g, b, y, mean, rstd, N, H * H, G, 1e-6, act, p, 1, None, ws))
    tp = timeit(lambda: call(lib, 'gn_fwd_pl_f32', x1, C1, x2, C2, g, b, None, planes, rec, mean, rstd, N, H * H, G, 1e-6, act, p, 1, None, ws)) \
        if lib.gn_fwd_pl_fused(C1, C2, H * H, G) else float('nan')
    tb = timeit(lambda: call(lib, 'gn_bwd_f32', dy, x1, C1, x2, C2, g, b, mean, rstd, dx1, 0.0, dx2, 0.0, dg, db, ws, N, H * H, G, act, p, 1, None))
    print(f'   act={act} p={p}: fwd {tf:6.1f} us ({2*nb/tf/1e3:5.0f} GB/s)  fwd->planes {tp:6.1f} us ({2*nb/tp/1e3:5.0f} GB/s)  '
          f'bwd {tb:6.1f} us ({3*nb/tb/1e3:5.0f} GB/s)')
