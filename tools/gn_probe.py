"""GroupNorm kernel probe: time fwd / bwd with and without SiLU and dropout (diagnostic)."""
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
for C, H in ((128, 32), (256, 16)):
  x = torch.randn(N, C, H, H, device=d); g = torch.ones(C, device=d); b = torch.zeros(C, device=d)
  y = torch.empty_like(x); mean = torch.empty(N * G, device=d); rstd = torch.empty(N * G, device=d)
  dy = torch.randn_like(x); dx = torch.empty_like(x); dg = torch.zeros(C, device=d); db = torch.zeros(C, device=d)
  ws = torch.empty(int(lib.gn_ws_bytes(N, C, H * H, G)) // 4 + 64, device=d)
  nb = x.numel() * 4
  for act, p in ((1, 0.1), (1, 0.0), (0, 0.0)):
    tf = timeit(lambda: call(lib, 'gn_fwd_f32', x, C, None, 0, g, b, y, mean, rstd, N, H * H, G, 1e-6, act, p, 1, None, ws))
    tb = timeit(lambda: call(lib, 'gn_bwd_f32', dy, x, C, None, 0, g, b, mean, rstd, dx, 0.0, None, 0.0, dg, db, ws, N, H * H, G, act, p, 1, None))
    print(f'C{C}@{H} act={act} p={p}: fwd {tf:6.1f} us ({2*nb/tf/1e3:5.0f} GB/s)  bwd {tb:6.1f} us ({3*nb/tb/1e3:5.0f} GB/s)')
