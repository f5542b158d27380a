#!/usr/bin/env python
"""Plane-operand convolution kernels with PREPARED weights (what the engine launches), one line per shape and direction:

    python tools/bench_x2d.py [--reps 50] [--tag name] [--batch B] [--shapes C1xHxCout,...]       (STK_LIBSTK=<other build> for an A/B of two library builds)

forward / data gradient (x2d::gemm_kernel, its K-split form on small maps) and the planes weight gradient, timed with HIP
events over `reps` back-to-back launches on random data; a checksum of every result is printed so that variants run in
separate processes (switches and STK_LIBSTK are read once per process) can be compared bit for bit.  Development tool."""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
  if p not in sys.path:
    sys.path.insert(0, p)

import numpy as np
import torch

import soft_truncation_amd as st
from _util import call


class Desc(ctypes.Structure):
  _fields_ = [('w', ctypes.c_void_p), ('wp', ctypes.c_void_p), ('sm', ctypes.c_long), ('sk', ctypes.c_long),
              ('M', ctypes.c_int), ('Kc', ctypes.c_int), ('Mpad', ctypes.c_int), ('taps', ctypes.c_int),
              ('flip', ctypes.c_int), ('reserved', ctypes.c_int)]


def timeit(fn, reps):
  for _ in range(5):
    fn()
  torch.cuda.synchronize()
  s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  s.record()
  for _ in range(reps):
    fn()
  e.record()
  torch.cuda.synchronize()
  return s.elapsed_time(e) * 1e3 / reps


def checksum(t):
  t = t.detach().double()
  return f'{float(t.sum()):+.10e}/{float(t.abs().max()):.8e}'


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--reps', type=int, default=50)
  ap.add_argument('--batch', type=int, default=128)
  ap.add_argument('--tag', default='default')
  ap.add_argument('--only', default='')
  ap.add_argument('--shapes', default='', help='comma-separated C1xHxCout 3x3 shapes instead of the built-in list, e.g. 256x32x256,512x8x256')
  args = ap.parse_args()
  lib = st.engine.lib.load()
  d = torch.device('cuda:0')
  N = args.batch
  shapes = [(128, 32, 128, 3), (256, 16, 256, 3), (384, 32, 128, 3), (512, 16, 256, 3), (128, 16, 256, 3),
            (256, 8, 256, 3), (512, 8, 256, 3), (256, 4, 256, 3), (512, 4, 256, 3), (256, 16, 768, 1)]
  if args.shapes:
    shapes = [tuple(int(v) for v in t.split('x')) + (3,) for t in args.shapes.split(',')]
  lines = []
  for C1, H, Cout, K in shapes:
    if args.only and f'{C1}x{H}' not in args.only.split(','):
      continue
    g = torch.Generator().manual_seed(C1 * 1000 + H)
    x = torch.randn(N, C1, H, H, generator=g).to(d)
    dy = torch.randn(N, Cout, H, H, generator=g).to(d)
    w = (torch.randn(Cout, C1, K, K, generator=g) / np.sqrt(C1 * K * K)).to(d)
    bias = torch.randn(Cout, generator=g).to(d)
    y = torch.empty(N, Cout, H, H, device=d)
    dx = torch.empty_like(x)
    dw = torch.zeros_like(w)
    shp = (C1, 0, N, H, H, Cout, K, K, 1, K // 2)
    if not int(lib.conv2d_pl_ok(0, C1, 0, N, H, H, Cout, K, K, 1, K // 2)):
      continue
    fb = max(int(lib.conv2d_fwd_ws_bytes(*shp)), int(lib.conv2d_dgrad_ws_bytes(*shp)), 1024)
    fws = torch.empty(fb // 4 + 64, device=d)
    blocks, descs, items = [], [], 0
    for direction in (0, 1):
      nb = int(lib.conv2d_wp_bytes(direction, *shp))
      blk = torch.zeros(nb + 256, dtype=torch.uint8, device=d)
      ptr = (blk.data_ptr() + 255) // 256 * 256
      desc = Desc()
      n = lib.conv2d_wp_desc(direction, w.data_ptr(), 0, C1, Cout, K, K, ptr, ctypes.byref(desc))
      assert n > 0
      items = max(items, n)
      descs.append(desc)
      blocks.append((blk, ptr))
    table = torch.from_numpy(np.frombuffer(b''.join(bytes(v) for v in descs), dtype=np.uint8).copy()).to(d)
    call(lib, 'conv2d_wprep_batch', table, len(descs), items)
    ax, ay = torch.zeros(256, device=d), torch.zeros(256, device=d)
    call(lib, 'amax_partial_f32', x, x.numel(), ax)
    call(lib, 'amax_partial_f32', dy, dy.numel(), ay)
    xp = torch.zeros(int(lib.planes_bytes(N, C1, H * H)), dtype=torch.uint8, device=d)
    yp = torch.zeros(int(lib.planes_bytes(N, Cout, H * H)), dtype=torch.uint8, device=d)
    call(lib, 'split_planes_f32', x, N, C1, H * H, ax, 256, xp)
    call(lib, 'split_planes_f32', dy, N, Cout, H * H, ay, 256, yp)
    flops = 2.0 * N * H * H * Cout * C1 * K * K
    shape = f'{C1}->{Cout} {K}x{K} @{H}x{H} b{N}'
    ks = int(lib.conv2d_pl_ksplit(0, C1, 0, N, H, H, Cout, K, K))

    def fwd():
      call(lib, 'conv2d_fwd_pl_f32', xp, ax, C1, w, 0, bias, None, 0, None, 1.0, y, N, H, H, Cout, K, K, blocks[0][1], fws, fb)

    def dgrad():
      call(lib, 'conv2d_dgrad_pl_f32', yp, ay, w, 0, dx, C1, 0.0, None, 0, 0.0, 1.0, N, H, H, Cout, K, K, blocks[1][1], fws, fb)

    for name, fn, out in (('fwd', fwd, y), ('dgrad', dgrad, dx)):
      us = timeit(fn, args.reps)
      lines.append(f'{args.tag:<8} {name:<6} {shape:<28} ksplit {ks:<2} {us:9.1f} us {flops / us / 1e6:7.1f} TF/s  {checksum(out)}')
      print(lines[-1], flush=True)
    if K == 3 and int(lib.conv2d_wgrad_pl_ok(N, H, H, C1, Cout)):
      nbp = int(lib.conv2d_wgrad_pl_ws_bytes(N, H, H, C1, Cout))
      wsp = torch.empty(nbp // 4 + 64, device=d)

      def wgrad():
        call(lib, 'conv2d_wgrad_pl_f32', xp, ax, yp, ay, dw, 1.0, wsp, nbp, N, H, H, C1, Cout)
      dw.zero_()
      wgrad()
      cs = checksum(dw)
      us = timeit(wgrad, args.reps)
      lines.append(f'{args.tag:<8} {"wgrad":<6} {shape:<28} slabs  {nbp // (4 * 9 * Cout * C1):<2} {us:9.1f} us {flops / us / 1e6:7.1f} TF/s  {cs}')
      print(lines[-1], flush=True)
  out = os.path.join(ROOT, 'gpurun_out', f'bench_x2d_{args.tag}.txt')
  os.makedirs(os.path.dirname(out), exist_ok=True)
  with open(out, 'w') as f:
    f.write('\n'.join(lines) + '\n')


if __name__ == '__main__':
  main()
