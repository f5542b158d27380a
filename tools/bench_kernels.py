#!/usr/bin/env python
"""Micro-benchmarks of individual C-ABI entry points on the shapes of the DDPM++ 32x32 / batch-128 step.

    python tools/bench_kernels.py [--only conv] [--reps 20]

Prints one line per (kernel, shape): average microseconds (HIP events on the launch stream), the
algorithmic TFLOP/s or GB/s and the fraction of the relevant MI355X roofline (157.3 TFLOP/s fp32 MFMA,
8.0 TB/s HBM).  Development tool: used to iterate on kernels between full bench.py runs.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
  if p not in sys.path:
    sys.path.insert(0, p)

import torch

import soft_truncation_amd as st
from _util import call


def timeit(fn, reps):
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  s.record()
  for _ in range(reps):
    fn()
  e.record()
  torch.cuda.synchronize()
  return s.elapsed_time(e) * 1e3 / reps   # us


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--only', default='')
  ap.add_argument('--reps', type=int, default=20)
  ap.add_argument('--batch', type=int, default=128)
  ap.add_argument('--shapes', type=int, default=0, help='only the first N conv shapes')
  ap.add_argument('--planes-only', action='store_true', help='conv section: time only the plane-operand calls')
  args = ap.parse_args()
  lib = st.engine.lib.load()
  d = torch.device('cuda:0')
  N = args.batch
  rows = []

  def rec(name, shape, us, flops=None, nbytes=None):
    if flops:
      tf = flops / us / 1e6
      rows.append(f'{name:<22} {shape:<34} {us:10.1f} us {tf:8.1f} TF/s  {tf / 157.3:6.1%} of fp32-MFMA')
    else:
      gb = nbytes / us / 1e3
      rows.append(f'{name:<22} {shape:<34} {us:10.1f} us {gb:8.0f} GB/s  {gb / 8000:6.1%} of HBM')
    print(rows[-1], flush=True)

  conv_shapes = [
    # C1, C2, H, Cout, K
    (128, 0, 32, 128, 3), (256, 0, 16, 256, 3), (256, 0, 8, 256, 3), (256, 0, 4, 256, 3),
    (128, 0, 16, 256, 3), (384, 0, 32, 128, 3), (512, 0, 16, 256, 3), (512, 0, 8, 256, 3),
    (256, 256, 16, 256, 1), (256, 0, 16, 256, 1), (128, 0, 16, 256, 1), (256, 128, 32, 128, 1),
  ]
  if 'hq' in args.only:        # the 256x256 network's layers (use with --batch 4)
    conv_shapes = [(128, 0, 256, 128, 3), (128, 0, 128, 128, 3), (128, 0, 64, 256, 3), (256, 0, 64, 256, 3),
                   (256, 0, 32, 256, 3), (256, 128, 256, 128, 3), (256, 0, 256, 128, 1)]
  if args.shapes:
    conv_shapes = conv_shapes[:args.shapes]
  if not args.only or 'conv' in args.only:
    for C1, C2, H, Cout, K in conv_shapes:
      if 'k1' in args.only and K != 1:
        continue
      Cin = C1 + C2
      x1 = torch.randn(N, C1, H, H, device=d)
      x2 = torch.randn(N, C2, H, H, device=d) if C2 else None
      w = torch.randn(Cout, Cin, K, K, device=d) * 0.02
      bias = torch.randn(Cout, device=d)
      y = torch.empty(N, Cout, H, H, device=d)
      dy = torch.randn(N, Cout, H, H, device=d)
      dx1 = torch.empty_like(x1)
      dx2 = torch.empty_like(x2) if C2 else None
      dw = torch.zeros_like(w)
      nb = int(lib.conv2d_wgrad_ws_bytes(C1, C2, N, Cout, H, H, K, K))
      ws = torch.empty(nb // 4 + 64, device=d)
      dims = (N, H, H, Cout, H, H, K, K, 1, K // 2)
      flops = 2.0 * N * H * H * Cout * Cin * K * K
      shape = f'{Cin}->{Cout} {K}x{K} @{H}x{H} b{N}' + (' dual' if C2 else '')
      shp = (C1, C2, N, H, H, Cout, K, K, 1, K // 2)
      fb = max(int(lib.conv2d_fwd_ws_bytes(*shp)), int(lib.conv2d_dgrad_ws_bytes(*shp)))
      fws = torch.empty(fb // 4 + 64, device=d)
      if not args.planes_only:
       rec('conv.fwd.f32in', shape, timeit(lambda: call(lib, 'conv2d_fwd_f32', x1, C1, x2, C2, w, 0, bias, None, 0, None, 1.0, y, *dims, None, 0), args.reps), flops)
      if not args.planes_only:
       rec('conv.dgrad.f32in', shape, timeit(lambda: call(lib, 'conv2d_dgrad_f32', dy, w, 0, dx1, C1, 0.0, dx2, C2, 0.0, 1.0, *dims, None, 0), args.reps), flops)
      if fb and not args.planes_only:
        rec('conv.fwd.split', shape, timeit(lambda: call(lib, 'conv2d_fwd_f32', x1, C1, x2, C2, w, 0, bias, None, 0, None, 1.0, y, *dims, fws, fb), args.reps), flops)
        rec('conv.dgrad.split', shape, timeit(lambda: call(lib, 'conv2d_dgrad_f32', dy, w, 0, dx1, C1, 0.0, dx2, C2, 0.0, 1.0, *dims, fws, fb), args.reps), flops)
      if fb and not C2 and int(lib.conv2d_pl_ok(0, C1, 0, N, H, H, Cout, K, K, 1, K // 2)):
        ax, ay = torch.zeros(256, device=d), torch.zeros(256, device=d)
        call(lib, 'amax_partial_f32', x1, x1.numel(), ax)
        call(lib, 'amax_partial_f32', dy, dy.numel(), ay)
        xp = torch.zeros(int(lib.planes_bytes(N, C1, H * H)), dtype=torch.uint8, device=d)
        yp = torch.zeros(int(lib.planes_bytes(N, Cout, H * H)), dtype=torch.uint8, device=d)
        rec('amax_partial', shape, timeit(lambda: call(lib, 'amax_partial_f32', x1, x1.numel(), ax), args.reps), nbytes=4 * x1.numel())
        rec('split_planes', shape, timeit(lambda: call(lib, 'split_planes_f32', x1, N, C1, H * H, ax, 256, xp), args.reps), nbytes=8 * x1.numel())
        call(lib, 'split_planes_f32', dy, N, Cout, H * H, ay, 256, yp)
        rec('conv.fwd.planes', shape, timeit(lambda: call(lib, 'conv2d_fwd_pl_f32', xp, ax, C1, w, 0, bias, None, 0, None, 1.0, y, N, H, H, Cout, K, K, None, fws, fb), args.reps), flops)
        rec('conv.dgrad.planes', shape, timeit(lambda: call(lib, 'conv2d_dgrad_pl_f32', yp, ay, w, 0, dx1, C1, 0.0, None, 0, 0.0, 1.0, N, H, H, Cout, K, K, None, fws, fb), args.reps), flops)
        if K == 3 and int(lib.conv2d_wgrad_pl_ok(N, H, H, C1, Cout)):
          nbp = int(lib.conv2d_wgrad_pl_ws_bytes(N, H, H, C1, Cout))
          wsp = torch.empty(nbp // 4 + 64, device=d)
          rec('conv.wgrad.planes', shape, timeit(lambda: call(lib, 'conv2d_wgrad_pl_f32', xp, ax, yp, ay, dw, 1.0, wsp, nbp, N, H, H, C1, Cout), args.reps), flops)
      if not args.planes_only:
       rec('conv.wgrad', shape, timeit(lambda: call(lib, 'conv2d_wgrad_f32', x1, C1, x2, C2, dy, dw, 0, 1.0, ws, ws.numel() * 4, *dims), args.reps), flops)

  if not args.only or 'gn' in args.only:
    for C, H, Nb in [(128, 32, N), (256, 16, N), (256, 8, N), (384, 32, N), (512, 16, N), (128, 256, 4), (256, 128, 4),
                     (128, 64, N), (256, 32, N), (384, 64, N), (512, 32, N), (256, 64, 4), (256, 256, 4)]:
      G = 32
      x = torch.randn(Nb, C, H, H, device=d)
      g, b = torch.ones(C, device=d), torch.zeros(C, device=d)
      y = torch.empty_like(x)
      mean, rstd = torch.empty(Nb * G, device=d), torch.empty(Nb * G, device=d)
      dy, dx = torch.randn_like(x), torch.empty_like(x)
      dg, db = torch.zeros(C, device=d), torch.zeros(C, device=d)
      ws = torch.empty(int(lib.gn_ws_bytes(Nb, C, H * H, G)) // 4 + 64, device=d)
      nb = x.numel() * 4
      shape = f'C{C} @{H}x{H} b{Nb}'
      rec('gn_silu.fwd', shape, timeit(lambda: call(lib, 'gn_fwd_f32', x, C, None, 0, g, b, y, mean, rstd, Nb, H * H, G, 1e-6, 1, 0.1, 1, None, ws), args.reps), nbytes=2 * nb)
      rec('gn_silu.bwd', shape, timeit(lambda: call(lib, 'gn_bwd_f32', dy, x, C, None, 0, g, b, mean, rstd, dx, 0.0, None, 0.0, dg, db, ws, Nb, H * H, G, 1, 0.1, 1, None), args.reps), nbytes=3 * nb)
      ax = torch.zeros(256, device=d)
      rec('amax_partial', shape, timeit(lambda: call(lib, 'amax_partial_f32', x, x.numel(), ax), args.reps), nbytes=nb)
      if C % 32 == 0:
        xp = torch.zeros(int(lib.planes_bytes(Nb, C, H * H)), dtype=torch.uint8, device=d)
        rec('split_planes', shape, timeit(lambda: call(lib, 'split_planes_f32', x, Nb, C, H * H, ax, 256, xp), args.reps), nbytes=2 * nb)
        del xp
      src, dst = torch.empty(3 * x.numel() // 2, device=d), torch.empty(3 * x.numel() // 2, device=d)
      rec('copy.like.gn_bwd', shape, timeit(lambda: dst.copy_(src), args.reps), nbytes=3 * nb)
      del src, dst, x, y, dy, dx

  if not args.only or 'attn' in args.only:
    # attention core: fused kernels against the GEMM + softmax sequence they replace (engine/graph.py AttentionCore)
    for C, T in ([(256, 256)] if 'attn256' in args.only else [(256, 256), (256, 64), (256, 16)]):
      B = N
      q, k, v, do = (torch.randn(B, C, T, device=d) for _ in range(4))
      o, dq, dk, dv = (torch.empty(B, C, T, device=d) for _ in range(4))
      lse, delta, rcd = torch.empty(B, T, device=d), torch.empty(B, T, device=d), torch.empty(1024, device=d)
      S, Pm = torch.empty(B, T, T, device=d), torch.empty(B, T, T, device=d)
      sc = float(C) ** -0.5
      shape = f'C{C} T{T} b{B}'
      fl = 2.0 * B * T * T * C
      rec('attn.fwd.fused', shape, timeit(lambda: call(lib, 'attention_fwd_f32', q, k, v, C * T, o, lse, rcd, B, C, T, sc), args.reps), flops=2 * fl)
      rec('attn.bwd.fused', shape, timeit(lambda: call(lib, 'attention_bwd_f32', q, k, v, C * T, do, lse, rcd, delta, dq, 0.0, dk, 0.0, dv, 0.0, C * T, B, C, T, sc), args.reps), flops=4 * fl)

      def unfused_fwd():
        call(lib, 'gemm_f32', q, 1, T, C * T, k, T, 1, C * T, S, T, 1, T * T, None, 0, T, T, C, B, 1.0, 0.0)
        call(lib, 'softmax_fwd_f32', S, Pm, B * T, T, sc)
        call(lib, 'gemm_f32', v, T, 1, C * T, Pm, 1, T, T * T, o, T, 1, C * T, None, 0, C, T, T, B, 1.0, 0.0)

      def unfused_bwd():
        call(lib, 'gemm_f32', do, 1, T, C * T, v, T, 1, C * T, S, T, 1, T * T, None, 0, T, T, C, B, 1.0, 0.0)
        call(lib, 'gemm_f32', do, T, 1, C * T, Pm, T, 1, T * T, dv, T, 1, C * T, None, 0, C, T, T, B, 1.0, 0.0)
        call(lib, 'softmax_bwd_f32', Pm, S, S, B * T, T, sc)
        call(lib, 'gemm_f32', k, T, 1, C * T, S, 1, T, T * T, dq, T, 1, C * T, None, 0, C, T, T, B, 1.0, 0.0)
        call(lib, 'gemm_f32', q, T, 1, C * T, S, T, 1, T * T, dk, T, 1, C * T, None, 0, C, T, T, B, 1.0, 0.0)
      if 'attn256' in args.only:
        continue
      rec('attn.fwd.gemms', shape, timeit(unfused_fwd, args.reps), flops=2 * fl)
      rec('attn.bwd.gemms', shape, timeit(unfused_bwd, args.reps), flops=4 * fl)

  if not args.only or 'bias' in args.only:
    for C, H in [(128, 32), (256, 16), (256, 8), (256, 4)]:
      dy = torch.randn(N, C, H, H, device=d)
      db, dt, ws = torch.zeros(C, device=d), torch.zeros(N, C, device=d), torch.empty(N * C + 64, device=d)
      shape = f'C{C} @{H}x{H} b{N}'
      rec('bias_grad', shape, timeit(lambda: call(lib, 'bias_grad_f32', dy, N, C, H * H, 1.0, None, 0, db, ws), args.reps), nbytes=dy.numel() * 4)
      rec('bias_grad+temb', shape, timeit(lambda: call(lib, 'bias_grad_f32', dy, N, C, H * H, 1.0, dt, C, db, ws), args.reps), nbytes=dy.numel() * 4)

  if not args.only or 'misc' in args.only:
    n = N * 256 * 16 * 16
    a, b2, o = torch.randn(n, device=d), torch.randn(n, device=d), torch.empty(n, device=d)
    rec('axpby', f'n={n}', timeit(lambda: call(lib, 'axpby_f32', a, 0.5, b2, 1.0, o, n), args.reps), nbytes=12 * n)
    planes, H = N * 256, 16
    x = torch.randn(planes, H, H, device=d)
    k = torch.ones(4, 4, device=d) / 16
    dn, up = torch.empty(planes, H // 2, H // 2, device=d), torch.empty(planes, 2 * H, 2 * H, device=d)
    rec('upfirdn2d.down', f'{planes}x{H}x{H}', timeit(lambda: call(lib, 'upfirdn2d_f32', x, k, dn, planes, H, H, 1, 4, 4, 1, 1, 2, 2, 1, 1, 1, 1), args.reps), nbytes=4 * (x.numel() + dn.numel()))
    rec('upfirdn2d.up', f'{planes}x{H}x{H}', timeit(lambda: call(lib, 'upfirdn2d_f32', x, k, up, planes, H, H, 1, 4, 4, 2, 2, 1, 1, 2, 1, 2, 1), args.reps), nbytes=4 * (x.numel() + up.numel()))
  if not args.only or 'upfirdn' in args.only:
    # the FIR resampling calls of the 64x64 / batch-128 and 256x256 / batch-4 nets (planes = batch x channels), beside a plain
    # device copy of the same number of bytes (what the HBM sustains for a read + write stream of that size)
    k = torch.ones(4, 4, device=d) / 16
    for planes, H in [(16384, 64), (32768, 32), (32768, 16), (512, 256), (512, 128), (1024, 64), (1024, 32), (1024, 16)]:
      x = torch.randn(planes, H, H, device=d)
      dn, up, same = torch.empty(planes, H // 2, H // 2, device=d), torch.empty(planes, 2 * H, 2 * H, device=d), torch.empty(planes, H + 1, H + 1, device=d)
      shape = f'{planes}x{H}x{H}'
      rec('upfirdn2d.down', shape, timeit(lambda: call(lib, 'upfirdn2d_f32', x, k, dn, planes, H, H, 1, 4, 4, 1, 1, 2, 2, 1, 1, 1, 1), args.reps), nbytes=4 * (x.numel() + dn.numel()))
      rec('upfirdn2d.up', shape, timeit(lambda: call(lib, 'upfirdn2d_f32', x, k, up, planes, H, H, 1, 4, 4, 2, 2, 1, 1, 2, 1, 2, 1), args.reps), nbytes=4 * (x.numel() + up.numel()))
      rec('upfirdn2d.fir', shape, timeit(lambda: call(lib, 'upfirdn2d_f32', x, k, same, planes, H, H, 1, 4, 4, 1, 1, 1, 1, 2, 2, 2, 2), args.reps), nbytes=4 * (x.numel() + same.numel()))
      for name, n_in, n_out in (('copy.like.down', x.numel(), dn.numel()), ('copy.like.up', x.numel(), up.numel())):
        n = (n_in + n_out) // 2
        src, dst = torch.empty(n, device=d), torch.empty(n, device=d)
        rec(name, shape, timeit(lambda: dst.copy_(src), args.reps), nbytes=8 * n)
      del x, dn, up, same

  if not args.only or 'misc' in args.only:
    P = 61804419
    p, g, m, v = (torch.randn(P, device=d) * 0.01 for _ in range(4))
    v = v.abs()
    ss, wsb = torch.zeros(1, device=d), torch.zeros(2048, device=d)
    rec('sumsq', f'n={P}', timeit(lambda: call(lib, 'sumsq_f32', g, P, ss, wsb), args.reps), nbytes=4 * P)
    rec('adam', f'n={P}', timeit(lambda: call(lib, 'adam_f32', p, g, m, v, P, 2e-4, 0.9, 0.999, 1e-8, 0.0, 0, 0.1, 0.001, ss, 1.0), args.reps), nbytes=28 * P)
    rec('ema', f'n={P}', timeit(lambda: call(lib, 'ema_f32', m, p, P, 1e-4), args.reps), nbytes=12 * P)

  out = os.path.join(ROOT, 'gpurun_out', 'bench_kernels.txt')
  os.makedirs(os.path.dirname(out), exist_ok=True)
  with open(out, 'w') as f:
    f.write('\n'.join(rows) + '\n')


if __name__ == '__main__':
  main()
