#!/usr/bin/env python
"""Do the data gradient (+ GroupNorm backward + planes split) and the weight gradient of a 3x3 layer overlap when they
run on two HIP streams?  Times K layers' worth of [split_planes, dgrad, gn_bwd] on one stream and [wgrad] on a second
one, against everything on one stream.  Development tool (decides whether the backward graph should fork)."""
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
from bench_x2d import Desc


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--reps', type=int, default=20)
  ap.add_argument('--batch', type=int, default=128)
  args = ap.parse_args()
  lib = st.engine.lib.load()
  d = torch.device('cuda:0')
  N = args.batch
  for C1, H, Cout in ((128, 32, 128), (256, 16, 256), (256, 8, 256)):
    K = 3
    g = torch.Generator().manual_seed(C1 * 1000 + H)
    x = torch.randn(N, C1, H, H, generator=g).to(d)
    dy = torch.randn(N, Cout, H, H, generator=g).to(d)
    w = (torch.randn(Cout, C1, K, K, generator=g) / np.sqrt(C1 * K * K)).to(d)
    dx = torch.empty_like(x)
    dx2 = torch.empty_like(x)
    dw = torch.zeros_like(w)
    shp = (C1, 0, N, H, H, Cout, K, K, 1, 1)
    fb = max(int(lib.conv2d_fwd_ws_bytes(*shp)), int(lib.conv2d_dgrad_ws_bytes(*shp)), 1024)
    fws = torch.empty(fb // 4 + 64, device=d)
    nb = int(lib.conv2d_wp_bytes(1, *shp))
    blk = torch.zeros(nb + 256, dtype=torch.uint8, device=d)
    ptr = (blk.data_ptr() + 255) // 256 * 256
    desc = Desc()
    n = lib.conv2d_wp_desc(1, w.data_ptr(), 0, C1, Cout, K, K, ptr, ctypes.byref(desc))
    table = torch.from_numpy(np.frombuffer(bytes(desc), dtype=np.uint8).copy()).to(d)
    call(lib, 'conv2d_wprep_batch', table, 1, n)
    ax, ay = torch.zeros(256, device=d), torch.zeros(256, device=d)
    call(lib, 'amax_partial_f32', x, x.numel(), ax)
    call(lib, 'amax_partial_f32', dy, dy.numel(), ay)
    xp = torch.zeros(int(lib.planes_bytes(N, C1, H * H)), dtype=torch.uint8, device=d)
    yp = torch.zeros(int(lib.planes_bytes(N, Cout, H * H)), dtype=torch.uint8, device=d)
    call(lib, 'split_planes_f32', x, N, C1, H * H, ax, 256, xp)
    nbp = int(lib.conv2d_wgrad_pl_ws_bytes(N, H, H, C1, Cout))
    wsp = torch.empty(nbp // 4 + 64, device=d)
    G = 32
    gamma, beta = torch.ones(C1, device=d), torch.zeros(C1, device=d)
    mean, rstd = torch.zeros(N * G, device=d), torch.ones(N * G, device=d)
    gws = torch.empty(int(lib.gn_ws_bytes(N, C1, H * H, G)) // 4 + 64, device=d)

    def chain():       # what the data-gradient chain of one convolution launches
      call(lib, 'split_planes_f32', dy, N, Cout, H * H, ay, 256, yp)
      call(lib, 'conv2d_dgrad_pl_f32', yp, ay, w, 0, dx, C1, 0.0, None, 0, 0.0, 1.0, N, H, H, Cout, K, K, ptr, fws, fb)
      call(lib, 'gn_bwd_f32', dx, x, C1, None, 0, gamma, beta, mean, rstd, dx2, 0.0, None, 0.0, None, None, gws, N, H * H, G, 1, 0.0,
           0, None)

    def wgrad():
      call(lib, 'conv2d_wgrad_pl_f32', xp, ax, yp, ay, dw, 1.0, wsp, nbp, N, H, H, C1, Cout)

    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def run(par):
      torch.cuda.synchronize()
      t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      t0.record()
      s1.wait_stream(torch.cuda.current_stream())
      s2.wait_stream(torch.cuda.current_stream())
      for _ in range(args.reps):
        with torch.cuda.stream(s1):
          chain()
        with torch.cuda.stream(s2 if par else s1):
          wgrad()
      torch.cuda.current_stream().wait_stream(s1)
      torch.cuda.current_stream().wait_stream(s2)
      t1.record()
      torch.cuda.synchronize()
      return t0.elapsed_time(t1) * 1e3 / args.reps

    def only(fn):
      torch.cuda.synchronize()
      t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      t0.record()
      for _ in range(args.reps):
        fn()
      t1.record()
      torch.cuda.synchronize()
      return t0.elapsed_time(t1) * 1e3 / args.reps

    run(False); run(True)
    print(f'{C1}->{Cout} @{H}x{H} b{N}: chain {only(chain):7.1f} us  wgrad {only(wgrad):7.1f} us  one stream {run(False):7.1f} us  '
          f'two streams {run(True):7.1f} us', flush=True)


if __name__ == '__main__':
  main()
