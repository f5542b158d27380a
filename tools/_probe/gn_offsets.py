"""Does the GroupNorm backward's bandwidth depend on the RELATIVE placement of x, dy and dx (HBM channel aliasing)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tools')):
  sys.path.insert(0, p)
import torch
import soft_truncation_amd as st
from _util import call
from bench_kernels import timeit
lib = st.engine.lib.load()
d = torch.device('cuda:0')
N, C, H, G = 128, 128, 32, 32
n = N * C * H * H
big = torch.randn(3 * n + 3 * (1 << 22), device=d)
g, b = torch.ones(C, device=d), torch.zeros(C, device=d)
mean, rstd = torch.zeros(N * G, device=d), torch.ones(N * G, device=d)
dg, db = torch.zeros(C, device=d), torch.zeros(C, device=d)
ws = torch.empty(int(lib.gn_ws_bytes(N, C, H * H, G)) // 4 + 64, device=d)
for delta in (0, 64, 256, 1024, 4096, 16384, 65536, 262144, 1048576 + 1024):   # floats
  x = big[0:n].view(N, C, H, H)
  dy = big[n + delta: 2 * n + delta].view(N, C, H, H)
  dx = big[2 * n + 2 * delta: 3 * n + 2 * delta].view(N, C, H, H)
  us = timeit(lambda: call(lib, 'gn_bwd_f32', dy, x, C, None, 0, g, b, mean, rstd, dx, 0.0, None, 0.0, dg, db, ws, N, H * H, G, 1, 0.0, 1, None), 30)
  print(f'skew {4 * delta:9d} B: {us:6.1f} us  {12 * n / us / 1e3:6.0f} GB/s', flush=True)
