"""GPU parity tests proper: every C-ABI entry of libstk.so (HIP, gfx950) against the oracle's plain-C
restatement (oracle/libstk_ref.so) on identical seeded inputs, called through the C ABI.

Tolerances: these are fp32 kernels checked against a double-accumulating restatement, so the bar is
"within fp32 round-off of the exact result": 1e-5 relative to the tensor's scale for streaming
kernels, 1e-4 for contractions with up to ~4.6k-term fp32 accumulation chains (the north-star
tolerance for end-to-end loss/score tensors is 1e-3; see test_gpu_model.py).
"""
import itertools

import numpy as np
import pytest
import torch

from _util import call, close, dev_of, rnd

pytestmark = pytest.mark.gpu


def both(ref_lib, hip_lib, fn):
  """Run `fn(lib, to)` on the checker and on the GPU; `to` moves a CPU tensor to the backend."""
  outs = []
  for lib in (ref_lib, hip_lib):
    d = dev_of(lib)
    outs.append(fn(lib, lambda t: None if t is None else t.to(d).contiguous()))
  if hip_lib.is_device:
    torch.cuda.synchronize()
  return outs


def compare(outs, rtol, what):
  ref, got = outs
  for k in ref:
    close(got[k], ref[k], rtol=rtol, what=f'{what}:{k}')


# ---------------------------------------------------------------------------------------------------
# upfirdn2d
# ---------------------------------------------------------------------------------------------------
FIR = np.outer([1, 3, 3, 1], [1, 3, 3, 1]).astype(np.float32) / 64.
UFD_CASES = [
  # (major, H, W, up, down, pad0, pad1, taps, beta)
  (6, 32, 32, 1, 2, 1, 1, FIR, 0.0),          # downsample_2d
  (6, 16, 16, 2, 1, 2, 1, FIR * 4, 0.0),      # upsample_2d
  (6, 32, 32, 1, 1, 2, 2, FIR, 0.0),          # conv_downsample_2d prefilter
  (6, 33, 33, 1, 1, 1, 1, FIR[::-1, ::-1].copy(), 0.5),   # its backward (flipped taps, pad 1,1), accumulate
  (5, 64, 64, 1, 2, 1, 1, FIR, 0.0),
  (3, 128, 128, 2, 1, 2, 1, FIR * 4, 0.0),
  (3, 256, 256, 1, 2, 1, 1, FIR, 0.0),
  (7, 8, 8, 1, 2, 1, 1, FIR, 0.0),            # tiny plane -> direct kernel
  (7, 4, 4, 2, 1, 2, 1, FIR * 4, 1.0),
  (4, 20, 24, 1, 2, 1, 1, FIR, 0.0),          # ragged sizes
  (4, 17, 19, 2, 1, 2, 1, FIR * 4, 0.0),
  (2, 40, 40, 3, 2, 2, 3, np.arange(25, dtype=np.float32).reshape(5, 5) / 25., 0.0),   # generic factors
  (2, 24, 24, 1, 1, -1, -1, FIR, 0.0),        # negative padding (crop)
  # rows up to 256 outputs wide are taken whole, in bands of rows (one contiguous run of input per workgroup)
  (3, 64, 64, 2, 1, 2, 1, FIR * 4, 0.0),      # 64 -> 128: bands of 32 rows
  (2, 128, 128, 2, 1, 2, 1, FIR * 4, 0.5),    # 128 -> 256
  (2, 128, 128, 1, 2, 1, 1, FIR, 0.5),        # 128 -> 64: bands of 32 rows
  (2, 128, 128, 1, 1, 2, 2, FIR, 0.0),        # 129 outputs per row in 256-wide bands
  (2, 129, 129, 1, 1, 1, 1, FIR[::-1, ::-1].copy(), 0.5),
  (2, 128, 64, 1, 2, 1, 1, FIR, 0.0),         # not square
  (2, 48, 200, 2, 1, 2, 1, FIR * 4, 0.0),     # 400 outputs per row: the 64-wide tiles
  (2, 70, 70, 2, 1, 2, 1, FIR * 4, 0.0),      # band whose run is not a whole number of 16-byte pieces
  (2, 100, 202, 1, 2, 1, 1, FIR, 1.0),        # ragged band: last band short, odd 16-byte alignment of the rows
  (1, 256, 256, 2, 1, 2, 1, FIR * 4, 0.0),    # 512 outputs per row: the 64-wide tiles
  # plain FIR with rows that are not a whole number of 16-byte pieces: the flat-order kernel
  (3, 64, 64, 1, 1, 2, 2, FIR, 0.0),          # 65 x 65, one plane per workgroup
  (5, 16, 16, 1, 1, 2, 2, FIR, 0.5),          # 17 x 17, several planes per workgroup, accumulate
  (9, 8, 8, 1, 1, 2, 2, FIR, 0.0),            # 9 x 9: more planes per workgroup than the last group holds
  (2, 30, 30, 1, 1, 2, 2, FIR, 1.0),          # input rows not in 16-byte pieces either
  (2, 40, 64, 1, 1, 2, 2, FIR, 0.0),          # 41 x 65
  (1, 256, 256, 1, 1, 2, 2, FIR, 0.0),        # 257 x 257: bands of 7 rows
]


@pytest.mark.parametrize('case', UFD_CASES, ids=lambda c: f'{c[0]}x{c[1]}x{c[2]}_u{c[3]}d{c[4]}p{c[5]}{c[6]}')
def test_upfirdn2d(ref_lib, hip_lib, case):
  major, H, W, up, down, p0, p1, taps, beta = case
  kh, kw = taps.shape
  oh = (H * up + p0 + p1 - kh) // down + 1
  ow = (W * up + p0 + p1 - kw) // down + 1
  x = rnd(major, H, W, 1, seed=1)
  k = torch.from_numpy(taps)
  o0 = rnd(major, oh, ow, 1, seed=2)

  def fn(lib, to):
    out = to(o0.clone())
    call(lib, 'upfirdn2d_acc_f32', to(x), to(k), out, float(beta), major, H, W, 1, kh, kw, up, up, down, down,
         p0, p1, p0, p1)
    out2 = to(torch.zeros_like(o0))
    call(lib, 'upfirdn2d_f32', to(x), to(k), out2, major, H, W, 1, kh, kw, up, up, down, down, p0, p1, p0, p1)
    return {'acc': out, 'plain': out2}

  compare(both(ref_lib, hip_lib, fn), 1e-5, 'upfirdn2d')


def test_upfirdn2d_minor_dim(ref_lib, hip_lib):
  x = rnd(3, 12, 12, 5, seed=3)
  k = torch.from_numpy(FIR)

  def fn(lib, to):
    out = to(torch.zeros(3, 6, 6, 5))
    call(lib, 'upfirdn2d_f32', to(x), to(k), out, 3, 12, 12, 5, 4, 4, 1, 1, 2, 2, 1, 1, 1, 1)
    return {'out': out}

  compare(both(ref_lib, hip_lib, fn), 1e-5, 'upfirdn2d.minor')


def test_upfirdn2d_adjoint_full_size(hip_lib):
  """Size-independent property at CelebA-64 size: <up(x), y> == <x, down_adjoint(y)> where the
  adjoint is the same operator with flipped taps, up<->down swapped and g_pad (op/upfirdn2d.py:111-114)."""
  d = dev_of(hip_lib)
  major, H = 128 * 16, 64
  x = rnd(major, H, H, 1, seed=4).to(d)
  k = torch.from_numpy(FIR).to(d)
  kf = torch.flip(k, [0, 1]).contiguous()
  y = torch.empty(major, H // 2, H // 2, 1, device=d)
  call(hip_lib, 'upfirdn2d_f32', x, k, y, major, H, H, 1, 4, 4, 1, 1, 2, 2, 1, 1, 1, 1)
  g = rnd(major, H // 2, H // 2, 1, seed=5).to(d)
  gx = torch.empty_like(x)
  # backward of (up1, down2, pad(1,1)): up2, down1, g_pad = (k-p0-1, in*up - out*down + p0 - up + 1) = (2, 1)
  call(hip_lib, 'upfirdn2d_f32', g, kf, gx, major, H // 2, H // 2, 1, 4, 4, 2, 2, 1, 1, 2, 1, 2, 1)
  lhs = (y.double() * g.double()).sum().item()
  rhs = (x.double() * gx.double()).sum().item()
  assert abs(lhs - rhs) <= 1e-6 * max(abs(lhs), abs(rhs), 1.0), (lhs, rhs)


# ---------------------------------------------------------------------------------------------------
# element-wise family
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('shape', [(2, 3, 5, 16), (3, 16, 16, 64), (1, 7, 2, 9)])
def test_concat(ref_lib, hip_lib, shape):
  """torch.cat along the channels, materialised, and its backward (Combine(method='cat'), models/layerspp.py:57-72)."""
  N, Ca, Cb, HW = shape
  a, b = rnd(N, Ca, HW, seed=1), rnd(N, Cb, HW, seed=2)
  dout = rnd(N, Ca + Cb, HW, seed=3)
  oa, ob = rnd(N, Ca, HW, seed=4), rnd(N, Cb, HW, seed=5)

  def fn(lib, to):
    y = to(torch.full((N, Ca + Cb, HW), float('nan')))
    call(lib, 'concat_f32', to(a), Ca, to(b), Cb, y, N, HW)
    da, db = to(oa.clone()), to(torch.full((N, Cb, HW), float('nan')))
    call(lib, 'concat_bwd_f32', to(dout), da, 0.5, Ca, db, 0.0, Cb, N, HW)
    return {'y': y, 'da': da, 'db': db}

  outs = both(ref_lib, hip_lib, fn)
  compare(outs, 1e-6, 'concat')
  assert torch.equal(outs[0]['y'].cpu(), torch.cat([a, b], dim=1)) and torch.equal(outs[0]['db'].cpu(), dout[:, Ca:])


@pytest.mark.parametrize('shape', [(2, 3, 16), (3, 3, 1024), (1, 5, 77)])
def test_fixed_fourier(ref_lib, hip_lib, shape):
  """layerspp.FixedFouriereProjection (models/layerspp.py:31-43) forward / backward; arguments reach 256 pi = 804 rad."""
  N, C, HW = shape
  x = torch.rand(N, C, HW, generator=torch.Generator().manual_seed(1)) * 2 - 1
  dy = rnd(N, 5 * C, HW, seed=2)
  old = rnd(N, C, HW, seed=3)

  def fn(lib, to):
    y = to(torch.full((N, 5 * C, HW), float('nan')))
    call(lib, 'fixed_fourier_fwd_f32', to(x), y, N, C, HW)
    dx = to(old.clone())
    call(lib, 'fixed_fourier_bwd_f32', to(x), to(dy), dx, 0.5, N, C, HW)
    dx0 = to(torch.full((N, C, HW), float('nan')))
    call(lib, 'fixed_fourier_bwd_f32', to(x), to(dy), dx0, 0.0, N, C, HW)
    return {'y': y, 'dx': dx, 'dx0': dx0}

  outs = both(ref_lib, hip_lib, fn)
  compare(outs, 1e-5, 'fixed fourier')
  want = torch.cat((x, torch.sin(x * 128 * np.pi), torch.cos(x * 128 * np.pi), torch.sin(x * 256 * np.pi), torch.cos(x * 256 * np.pi)), dim=1)
  assert (outs[0]['y'].cpu() - want).abs().max().item() <= 2e-6      # the checker against the torch expression itself



@pytest.mark.parametrize('n', [1, 7, 1024, 4099, 1 << 20])
def test_elementwise(ref_lib, hip_lib, n):
  a, b, g = rnd(n, seed=1), rnd(n, seed=2), rnd(n, seed=3)

  def fn(lib, to):
    o = {}
    y = to(torch.zeros(n)); call(lib, 'silu_fwd_f32', to(a), y, n); o['silu'] = y
    dx = to(b.clone()); call(lib, 'silu_bwd_f32', to(a), to(g), dx, 0.5, n); o['silu_bwd'] = dx
    dx0 = to(torch.full((n,), float('nan'))); call(lib, 'silu_bwd_f32', to(a), to(g), dx0, 0.0, n); o['silu_bwd0'] = dx0
    for code in (0, 1, 2, 3, 4):       # layers.get_act by code (include/stk.h STK_ACT_*)
      y = to(torch.zeros(n)); call(lib, 'act_fwd_f32', to(a), y, n, code); o[f'act{code}'] = y
      dx = to(b.clone()); call(lib, 'act_bwd_f32', to(a), to(g), dx, 0.5, n, code); o[f'act{code}_bwd'] = dx
    z = to(torch.zeros(n)); call(lib, 'axpby_f32', to(a), 1.5, to(b), -0.25, z, n); o['axpby'] = z
    z2 = to(torch.zeros(n)); call(lib, 'axpby_f32', to(a), 0.7, None, 0.0, z2, n); o['ax'] = z2
    z3 = to(b.clone()); call(lib, 'axpby_f32', to(a), 0.7, z3, 1.0, z3, n); o['axpby_alias'] = z3
    # beta == 0: the accumulated operand must not be read (0 * NaN would poison the result)
    z4 = to(torch.full((n,), float('nan'))); call(lib, 'axpby_f32', to(a), 0.7, z4, 0.0, z4, n); o['axpby_beta0'] = z4
    w = to(torch.zeros(n)); call(lib, 'add_div_f32', to(a), to(b), float(np.float32(np.sqrt(2.))), w, n); o['add_div'] = w
    w1 = to(torch.zeros(n)); call(lib, 'add_div_f32', to(a), to(b), 1.0, w1, n); o['add'] = w1
    v = to(torch.zeros(n)); call(lib, 'affine_f32', to(a), 2.0, -1.0, v, n); o['affine'] = v
    m = to(torch.zeros(n)); call(lib, 'dropout_mask_f32', m, n, 0.1, 12345); o['mask'] = m
    return o

  outs = both(ref_lib, hip_lib, fn)
  compare(outs, 2e-6, 'elementwise')
  assert torch.equal(outs[0]['mask'], outs[1]['mask'].cpu()), 'dropout mask must be bit-identical across backends'


@pytest.mark.parametrize('shape,act,grad', [((4, 6, 8, 8), 3, 0), ((4, 6, 8, 8), 3, 1), ((5, 7), 3, 0),
                                            ((3, 5, 9), 1, 0), ((2, 4, 4, 4), 3, 2)])
def test_fused_bias_act(ref_lib, hip_lib, shape, act, grad):
  x, ref = rnd(*shape, seed=1), rnd(*shape, seed=2)
  bias = rnd(shape[1], seed=3)
  n = x.numel()
  step_b = int(np.prod(shape[2:])) if len(shape) > 2 else 1

  def fn(lib, to):
    o = to(torch.zeros(shape))
    call(lib, 'fused_bias_act_f32', to(x), to(bias), to(ref) if grad == 1 else None, o, n, step_b, shape[1], act, grad,
         0.2, float(2 ** 0.5))
    o2 = to(torch.zeros(shape))
    call(lib, 'fused_bias_act_f32', to(x), None, to(ref), o2, n, 1, 1, act, grad, 0.1, 1.0)
    return {'bias': o, 'nobias': o2}

  compare(both(ref_lib, hip_lib, fn), 1e-6, 'fused_bias_act')


@pytest.mark.parametrize('planes,H,W', [(6, 16, 16), (3, 32, 32), (5, 4, 4), (2, 6, 10)])
def test_resample_naive(ref_lib, hip_lib, planes, H, W):
  x = rnd(planes, H, W, seed=1)
  up0, dn0 = rnd(planes, 2 * H, 2 * W, seed=2), rnd(planes, H // 2, W // 2, seed=3)

  def fn(lib, to):
    u = to(up0.clone()); call(lib, 'resample_naive_f32', to(x), u, planes, H, W, 0, 1.0, 0.0)
    ub = to(up0.clone()); call(lib, 'resample_naive_f32', to(x), ub, planes, H, W, 0, 0.25, 1.0)
    d = to(dn0.clone()); call(lib, 'resample_naive_f32', to(x), d, planes, H, W, 1, 1.0, 0.0)
    db = to(dn0.clone()); call(lib, 'resample_naive_f32', to(x), db, planes, H, W, 1, 4.0, 1.0)
    return {'up': u, 'up_acc': ub, 'down': d, 'down_acc': db}

  compare(both(ref_lib, hip_lib, fn), 1e-6, 'resample')


def test_small_helpers(ref_lib, hip_lib):
  B, inner = 6, 3 * 16 * 16
  x, z = rnd(B, inner, seed=1), rnd(B, inner, seed=2)
  s = torch.rand(B, generator=torch.Generator().manual_seed(3)) + 0.5
  a = torch.rand(B, generator=torch.Generator().manual_seed(4))
  t = torch.rand(B, generator=torch.Generator().manual_seed(5)) * 999
  freqs = torch.exp(torch.arange(64, dtype=torch.float32) * -(np.log(10000) / 63))
  W = rnd(32, seed=6, scale=16.)
  ls = torch.log(s)
  wgt = torch.rand(B, generator=torch.Generator().manual_seed(7)) * 10
  dl = rnd(B, seed=8)

  def fn(lib, to):
    o = {}
    r = to(torch.zeros(B, inner)); call(lib, 'rowscale_f32', to(x), to(s), r, B, inner, 1); o['rowdiv'] = r
    r2 = to(torch.zeros(B, inner)); call(lib, 'rowscale_f32', to(x), to(s), r2, B, inner, 0); o['rowmul'] = r2
    p = to(torch.zeros(B, inner)); call(lib, 'perturb_f32', to(x), to(z), to(a), to(s), p, B, inner); o['perturb'] = p
    p2 = to(torch.zeros(B, inner)); call(lib, 'perturb_f32', to(x), to(z), None, to(s), p2, B, inner); o['perturb_ve'] = p2
    e = to(torch.zeros(B, 128)); call(lib, 'timestep_embedding_f32', to(t), to(freqs), e, B, 128); o['temb'] = e
    f = to(torch.zeros(B, 64)); call(lib, 'fourier_embedding_f32', to(ls), to(W), f, B, 32); o['fourier'] = f
    for vp, mode, rm in itertools.product((0, 1), (0, 1), (0, 1)):
      l = to(torch.zeros(B)); call(lib, 'sm_loss_fwd_f32', to(x), to(z), to(s), to(wgt), l, B, inner, vp, mode, rm)
      o[f'loss{vp}{mode}{rm}'] = l
      g = to(torch.zeros(B, inner))
      call(lib, 'sm_loss_bwd_f32', to(x), to(z), to(s), to(wgt), to(dl), g, B, inner, vp, mode, rm)
      o[f'dloss{vp}{mode}{rm}'] = g
    return o

  outs = both(ref_lib, hip_lib, fn)
  ref, got = outs
  for k in ref:
    # sin/cos of arguments up to ~1e3 (temb) / ~1e2 (fourier): device and host libm agree to ~1 ulp of the result
    close(got[k], ref[k], rtol=2e-5 if k in ('temb', 'fourier') else 2e-6, what=k)


# ---------------------------------------------------------------------------------------------------
# GroupNorm (+SiLU) (+dropout)
# ---------------------------------------------------------------------------------------------------
GN_CASES = [
  # N, C1, C2, H, G, act, p
  (3, 128, 0, 8, 32, 1, 0.0),
  (2, 256, 128, 8, 32, 1, 0.0),      # concat, groups of 12 straddle the two sources
  (2, 128, 256, 4, 32, 1, 0.1),
  (3, 16, 0, 16, 4, 0, 0.0),         # attention's GroupNorm (no activation)
  (2, 32, 16, 16, 12, 1, 0.2),
  (2, 128, 0, 32, 32, 1, 0.1),
  (1, 256, 256, 16, 32, 1, 0.0),
  (2, 12, 0, 5, 3, 1, 0.0),          # HW = 25: scalar path
  (1, 128, 0, 64, 32, 1, 0.0),
  (2, 64, 0, 128, 32, 1, 0.1),       # 32 K-element groups: split over workgroups (HW = 4 chunks)
  (1, 32, 32, 64, 8, 1, 0.0),        # split path, concat input, 8 channels per group
  # the other activations of layers.get_act (models/layers.py:29-41): ReLU, LeakyReLU(0.2), ELU -- flat, concat and split paths
  (3, 128, 0, 8, 32, 2, 0.0), (2, 256, 128, 8, 32, 3, 0.1), (2, 128, 0, 32, 32, 4, 0.0), (2, 64, 0, 128, 32, 4, 0.1),
  (1, 32, 32, 64, 8, 3, 0.0), (2, 12, 0, 5, 3, 2, 0.0),
]


@pytest.mark.parametrize('case', GN_CASES, ids=str)
def test_groupnorm(ref_lib, hip_lib, case):
  N, C1, C2, H, G, act, p = case
  C, HW = C1 + C2, H * H
  x1 = rnd(N, C1, H, H, seed=1) * 2 + 0.5
  x2 = rnd(N, C2, H, H, seed=2) - 0.3 if C2 else None
  gamma, beta = rnd(C, seed=3) + 1, rnd(C, seed=4)
  dy = rnd(N, C, H, H, seed=5)
  d1, d2 = rnd(N, C1, H, H, seed=6), (rnd(N, C2, H, H, seed=7) if C2 else None)
  seed = 0xABCDEF12345

  def fn(lib, to):
    dev = dev_of(lib)
    y, mean, rstd = to(torch.zeros(N, C, H, H)), to(torch.zeros(N * G)), to(torch.zeros(N * G))
    sdev = torch.tensor([17], dtype=torch.int64, device=dev)
    a1, a2, ga, be = to(x1), to(x2), to(gamma), to(beta)
    ws = to(torch.zeros(max(int(lib.gn_ws_bytes(N, C, HW, G)) // 4, 2 * N * C)))
    call(lib, 'gn_fwd_f32', a1, C1, a2, C2, ga, be, y, mean, rstd, N, HW, G, 1e-6, act, p, seed, sdev, ws)
    dx1, dx2 = to(d1.clone()), to(d2.clone()) if C2 else None
    dg, db = to(torch.ones(C)), to(torch.ones(C))
    call(lib, 'gn_bwd_f32', to(dy), a1, C1, a2, C2, ga, be, mean, rstd, dx1, 0.0, dx2, 1.0, dg, db, ws, N, HW, G, act,
         p, seed, sdev)
    o = {'y': y, 'mean': mean, 'rstd': rstd, 'dx1': dx1, 'dgamma': dg, 'dbeta': db}
    if C2:
      o['dx2'] = dx2
    return o

  compare(both(ref_lib, hip_lib, fn), 2e-5, 'groupnorm')


GN_OUT_CASES = [
  # N, C1, C2, H, G, act, p, beta1, add
  (4, 128, 0, 32, 32, 1, 0.1, 0.0, False),    # the 32x32 layers: IPT 4
  (3, 256, 0, 16, 32, 1, 0.0, 0.0, True),     # + the identity skip's gradient added on the way
  (5, 256, 0, 4, 32, 1, 0.1, 1.0, True),      # 4x4 maps, accumulating into dx1: the by-products come from the FINAL values
  (2, 96, 0, 8, 24, 0, 0.0, 0.0, False),
  (3, 128, 128, 16, 32, 1, 0.1, 1.0, False),  # two sources (up path): by-products of dx1 only
  (2, 256, 128, 8, 32, 1, 0.0, 1.0, True),    # groups of 12 straddle the two sources
]


@pytest.mark.parametrize('case', GN_OUT_CASES, ids=str)
def test_groupnorm_backward_byproducts(ref_lib, hip_lib, case):
  """stk_gn_bwd_out_f32: dx1 / dx2 as stk_gn_bwd_f32 (+ the added branch), plus the per-(sample, channel) sums of dx1
  (fold-slot format and the strided time-embedding gradient) and the atomic-maximum scale record, against the oracle."""
  N, C1, C2, H, G, act, p, b1, with_add = case
  C, HW = C1 + C2, H * H
  x1 = rnd(N, C1, H, H, seed=1) * 2 + 0.5
  x2 = rnd(N, C2, H, H, seed=2) - 0.3 if C2 else None
  gamma, beta = rnd(C, seed=3) + 1, rnd(C, seed=4)
  dy = rnd(N, C, H, H, seed=5) * torch.logspace(-3, 0, N)[:, None, None, None]
  d1, d2 = rnd(N, C1, H, H, seed=6), (rnd(N, C2, H, H, seed=7) if C2 else None)
  add = rnd(N, C1, H, H, seed=8) if with_add else None
  seed, stride, col = 0xABCDEF12345, 2 * C1 + 5, 3
  assert int(hip_lib.gn_bwd_out_ok(C1, C2, HW, G)) == 1 and int(ref_lib.gn_bwd_out_ok(C1, C2, HW, G)) == 1

  def fn(lib, to):
    dev = dev_of(lib)
    y, mean, rstd = to(torch.zeros(N, C, H, H)), to(torch.zeros(N * G)), to(torch.zeros(N * G))
    sdev = torch.tensor([17], dtype=torch.int64, device=dev)
    a1, a2, ga, be = to(x1), to(x2), to(gamma), to(beta)
    ws = to(torch.zeros(max(int(lib.gn_ws_bytes(N, C, HW, G)) // 4, 2 * N * C)))
    call(lib, 'gn_fwd_f32', a1, C1, a2, C2, ga, be, y, mean, rstd, N, HW, G, 1e-6, act, p, seed, sdev, ws)
    dx1, dx2 = to(d1.clone()), (to(d2.clone()) if C2 else None)
    dsum, dtemb, rec = to(torch.full((N, C1, 2), 7.0)), to(torch.full((N * stride,), 7.0)), to(torch.zeros(256))
    call(lib, 'gn_bwd_out_f32', to(dy), a1, C1, a2, C2, ga, be, mean, rstd, dx1, b1, dx2, 1.0, None, None, ws, N, HW, G, act, p,
         seed, sdev, to(add), 0.75, dsum, 0.5, dtemb[col:], stride, rec)
    dtemb = dtemb.view(N, stride)
    o = {'dx1': dx1, 'sum': dsum[:, :, 0].contiguous(), 'sum2': dsum[:, :, 1].contiguous(),
         'temb': dtemb[:, col:col + C1].contiguous(), 'untouched': dtemb[:, :col].contiguous(),
         'amax': rec.max().reshape(1), 'fold': ws[:2 * N * C].clone()}
    if C2:
      o['dx2'] = dx2
    return o

  out = both(ref_lib, hip_lib, fn)
  compare(out, 2e-5, 'groupnorm by-products')
  got = out[1]
  assert float(got['amax']) == float(got['dx1'].abs().max())            # the record is exact, not approximate


# ---------------------------------------------------------------------------------------------------
# convolution: forward / dgrad / wgrad / bias-grad
# ---------------------------------------------------------------------------------------------------
CONV_CASES = [
  # N, C1, C2, H, W, Cout, K, stride, pad, OH, OW, layout, temb, res, div
  (2, 16, 0, 16, 16, 32, 3, 1, 1, 16, 16, 0, True, True, True),
  (4, 128, 0, 32, 32, 128, 3, 1, 1, 32, 32, 0, True, False, False),     # 128x128 tiles
  (3, 96, 64, 16, 16, 128, 3, 1, 1, 16, 16, 0, False, True, True),      # concat input
  (2, 3, 0, 32, 32, 128, 3, 1, 1, 32, 32, 0, False, False, False),      # stem
  (2, 128, 0, 32, 32, 3, 3, 1, 1, 32, 32, 0, False, False, False),      # head
  (3, 64, 0, 17, 17, 48, 3, 2, 0, 8, 8, 0, False, True, True),          # conv_downsample_2d's strided conv
  (2, 32, 0, 16, 16, 32, 3, 2, 0, 8, 8, 0, False, False, False),        # F.pad(0,1,0,1) + stride 2 (asymmetric)
  (3, 64, 32, 8, 8, 64, 1, 1, 0, 8, 8, 0, False, False, False),         # 1x1 shortcut on a concat
  (3, 256, 0, 16, 16, 256, 1, 1, 0, 16, 16, 1, False, True, True),      # NIN_3 + residual
  (5, 40, 0, 4, 4, 56, 1, 1, 0, 4, 4, 1, False, False, False),          # NIN, ragged channels
  (2, 20, 12, 12, 10, 24, 3, 1, 1, 12, 10, 0, True, True, False),       # ragged everything
  (130, 128, 0, 4, 4, 128, 3, 1, 1, 4, 4, 0, True, False, True),        # many tiny images per tile
  (24, 128, 0, 32, 32, 128, 3, 1, 1, 32, 32, 0, True, True, True),      # >= 192 tiles: fp16 two-way-split kernel
  (48, 64, 96, 16, 16, 160, 3, 1, 1, 16, 16, 0, False, True, False),    # split kernel: concat input, ragged Cout
  (43, 96, 0, 24, 24, 128, 3, 1, 1, 24, 24, 0, True, False, False),     # split kernel: ragged pixel tiles
  (6, 64, 0, 8, 8, 96, 3, 1, 1, 8, 8, 0, False, False, False),          # split wgrad: 8-wide maps (runs span two rows)
  (3, 32, 64, 8, 8, 72, 3, 1, 1, 8, 8, 0, False, False, False),         # split wgrad: 8-wide, concat, ragged Cout
  (48, 256, 0, 16, 16, 256, 1, 1, 0, 16, 16, 1, False, True, True),     # split kernel: NIN (w[Cin][Cout]) + residual
  (48, 128, 128, 16, 16, 256, 1, 1, 0, 16, 16, 0, False, False, False), # split kernel: 1x1 shortcut on a concat
  (96, 128, 0, 16, 16, 160, 1, 1, 0, 16, 16, 1, True, False, False),    # split kernel: NIN, ragged Cout
  (3, 3, 0, 16, 16, 40, 1, 1, 0, 16, 16, 0, False, True, False),        # thin input, 1x1 (input_skip Combine)
  (2, 64, 32, 8, 8, 3, 3, 1, 1, 8, 8, 0, False, True, True),            # thin output on a concat input (output_skip)
  (2, 160, 0, 12, 20, 2, 3, 1, 1, 12, 20, 0, True, False, False),       # thin output, > 128 input channels, ragged map
  (40, 64, 64, 8, 8, 200, 3, 1, 1, 8, 8, 0, True, True, True),          # split kernel, K split into slabs: concat, ragged
  (100, 256, 0, 4, 4, 256, 3, 1, 1, 4, 4, 0, True, True, False),        # K-split slabs on 4x4 maps
  (4, 256, 0, 4, 4, 256, 3, 1, 1, 4, 4, 0, True, False, True),          # K-split, one half-empty pixel tile (batch 4)
  (4, 128, 128, 8, 8, 256, 3, 1, 1, 8, 8, 0, False, True, False),       # K-split, 4 tiles, concat input (batch 4)
]


def _conv_id(c):
  return f'N{c[0]}_C{c[1]}+{c[2]}_{c[3]}x{c[4]}_o{c[5]}_k{c[6]}s{c[7]}p{c[8]}_l{c[11]}'


@pytest.mark.parametrize('scratch', [True, False], ids=['ws', 'nows'])
@pytest.mark.parametrize('case', CONV_CASES, ids=_conv_id)
def test_conv(ref_lib, hip_lib, case, scratch):
  """scratch=True hands fwd / dgrad their workspace (shapes that qualify then take the fp16 two-way-split
  kernel); scratch=False keeps every shape on the f32-input MFMA kernels.  Same tolerance for both."""
  N, C1, C2, H, W, Cout, K, stride, pad, OH, OW, layout, use_temb, use_res, use_div = case
  Cin = C1 + C2
  x1 = rnd(N, C1, H, W, seed=1)
  x2 = rnd(N, C2, H, W, seed=2) if C2 else None
  w = (rnd(Cout, Cin, K, K, seed=3) if layout == 0 else rnd(Cin, Cout, seed=3)) * (1.0 / np.sqrt(Cin * K * K))
  bias = rnd(Cout, seed=4)
  TS = Cout + 24
  temb = rnd(N, TS, seed=5) if use_temb else None
  res = rnd(N, Cout, OH, OW, seed=6) if use_res else None
  div = float(np.float32(np.sqrt(2.))) if use_div else 1.0
  dy = rnd(N, Cout, OH, OW, seed=7)
  g1, g2 = rnd(N, C1, H, W, seed=8), (rnd(N, C2, H, W, seed=9) if C2 else None)
  dw0, db0 = rnd(*w.shape, seed=10), rnd(Cout, seed=11)
  dims = (N, H, W, Cout, OH, OW, K, K, stride, pad)

  def fn(lib, to):
    o = {}
    a1, a2, ww = to(x1), to(x2), to(w)
    tp = None
    if use_temb:
      tt = to(temb)
      tp = tt.data_ptr() + 4 * 8            # a column slice of a wider [N, TS] tensor
    shape = (C1, C2, N, H, W, Cout, K, K, stride, pad)
    fb = max(int(lib.conv2d_fwd_ws_bytes(*shape)), int(lib.conv2d_dgrad_ws_bytes(*shape))) if scratch else 0
    fws = to(torch.zeros(fb // 4 + 64)) if fb else None
    y = to(torch.zeros(N, Cout, OH, OW))
    call(lib, 'conv2d_fwd_f32', a1, C1, a2, C2, ww, layout, to(bias), tp, TS if use_temb else 0, to(res), div, y, *dims,
         fws, fb)
    o['y'] = y
    y2 = to(torch.zeros(N, Cout, OH, OW))
    call(lib, 'conv2d_fwd_f32', a1, C1, a2, C2, ww, layout, None, None, 0, None, 1.0, y2, *dims, fws, fb)
    o['y_plain'] = y2
    d = to(dy)
    dx1, dx2 = to(g1.clone()), (to(g2.clone()) if C2 else None)
    call(lib, 'conv2d_dgrad_f32', d, ww, layout, dx1, C1, 0.0, dx2, C2, 1.0, 0.5, *dims, fws, fb)
    o['dx1'] = dx1
    if C2:
      o['dx2'] = dx2
    nbytes = int(lib.conv2d_wgrad_ws_bytes(C1, C2, N, Cout, OH, OW, K, K))
    ws = to(torch.zeros(max(nbytes // 4, N * Cout, 64)))
    dw = to(dw0.clone())
    call(lib, 'conv2d_wgrad_f32', a1, C1, a2, C2, d, dw, layout, 0.5, ws, ws.numel() * 4, *dims)
    o['dw'] = dw
    db = to(db0.clone())
    dt = to(torch.zeros(N, TS))
    call(lib, 'bias_grad_f32', d, N, Cout, OH * OW, 0.5, dt.data_ptr() + 4 * 8, TS, db, ws)
    o['dbias'], o['dtemb'] = db, dt
    db2 = to(db0.clone())
    call(lib, 'bias_grad_f32', d, N, Cout, OH * OW, 1.0, None, 0, db2, ws)
    o['dbias_only'] = db2
    return o

  compare(both(ref_lib, hip_lib, fn), 1e-4, 'conv')


WP_CASES = [c for c in CONV_CASES if c[7] == 1 and c[3] == c[9] and c[4] == c[10]]


@pytest.mark.parametrize('case', WP_CASES, ids=_conv_id)
def test_conv_prepared_weights(hip_lib, case):
  """include/stk.h "Prepared weights": one batched launch prepares the forward and data-gradient blocks of a layer;
  the _wp calls must then be BIT-identical to the calls that prepare into their scratch (same kernels, same
  operands).  Shapes that do not run on the split kernel report 0 bytes and reject a block."""
  import ctypes

  class Desc(ctypes.Structure):
    _fields_ = [('w', ctypes.c_void_p), ('wp', ctypes.c_void_p), ('sm', ctypes.c_long), ('sk', ctypes.c_long),
                ('M', ctypes.c_int), ('Kc', ctypes.c_int), ('Mpad', ctypes.c_int), ('taps', ctypes.c_int),
                ('flip', ctypes.c_int), ('reserved', ctypes.c_int)]

  N, C1, C2, H, W, Cout, K, stride, pad, OH, OW, layout, use_temb, use_res, use_div = case
  Cin = C1 + C2
  d = dev_of(hip_lib)
  lib = hip_lib
  x1 = rnd(N, C1, H, W, seed=1).to(d)
  x2 = rnd(N, C2, H, W, seed=2).to(d) if C2 else None
  w = ((rnd(Cout, Cin, K, K, seed=3) if layout == 0 else rnd(Cin, Cout, seed=3)) * (1.0 / np.sqrt(Cin * K * K))).to(d)
  bias, dy = rnd(Cout, seed=4).to(d), rnd(N, Cout, OH, OW, seed=7).to(d)
  dims = (N, H, W, Cout, OH, OW, K, K, stride, pad)
  shape = (C1, C2, N, H, W, Cout, K, K, stride, pad)
  fb = max(int(lib.conv2d_fwd_ws_bytes(*shape)), int(lib.conv2d_dgrad_ws_bytes(*shape)))
  nb = [int(lib.conv2d_wp_bytes(direction, *shape)) for direction in (0, 1)]
  assert (nb[0] > 0) == (int(lib.conv2d_variant(0, C1, C2, N, H, W, Cout, OH, OW, K, K, stride, pad, layout)) in (2, 5))
  assert (nb[1] > 0) == (int(lib.conv2d_variant(1, C1, C2, N, H, W, Cout, OH, OW, K, K, stride, pad, layout)) in (2, 5))
  fws = torch.full((fb // 4 + 64,), float('nan'), device=d)
  blocks, descs, items = [], [], 0
  for direction in (0, 1):
    if nb[direction] == 0:
      blocks.append(None)
      continue
    blk = torch.full((nb[direction] + 256,), 0xff, dtype=torch.uint8, device=d)
    ptr = (blk.data_ptr() + 255) // 256 * 256
    desc = Desc()
    n = lib.conv2d_wp_desc(direction, w.data_ptr(), layout, Cin, Cout, K, K, ptr, ctypes.byref(desc))
    assert n > 0
    items = max(items, n)
    descs.append(desc)
    blocks.append((blk, ptr))
  if descs:
    table = torch.from_numpy(np.frombuffer(b''.join(bytes(x) for x in descs), dtype=np.uint8).copy()).to(d)
    call(lib, 'conv2d_wprep_batch', table, len(descs), items)
  amax = torch.full((768,), float('nan'), device=d)       # the layer's amax buffer (include/stk.h)
  y_a, y_b = torch.empty(N, Cout, OH, OW, device=d), torch.empty(N, Cout, OH, OW, device=d)
  call(lib, 'conv2d_fwd_f32', x1, C1, x2, C2, w, layout, bias, None, 0, None, 1.0, y_a, *dims, fws, fb)
  dx_a = [torch.zeros(N, C1, H, W, device=d), torch.zeros(N, C2, H, W, device=d) if C2 else None]
  dx_b = [torch.zeros(N, C1, H, W, device=d), torch.zeros(N, C2, H, W, device=d) if C2 else None]
  call(lib, 'conv2d_dgrad_f32', dy, w, layout, dx_a[0], C1, 0.0, dx_a[1], C2, 0.0, 1.0, *dims, fws, fb)
  if blocks[0] is not None:
    call(lib, 'conv2d_fwd_wp_f32', x1, C1, x2, C2, w, layout, bias, None, 0, None, 1.0, y_b, *dims, blocks[0][1], amax, fws, fb)
    assert torch.equal(y_a, y_b)
  else:          # a block for a shape that has none is an error, not silently ignored ... unless a streaming kernel took it
    if int(lib.conv2d_variant(0, C1, C2, N, H, W, Cout, OH, OW, K, K, stride, pad, layout)) != 4:
      rc = lib.conv2d_fwd_wp_f32.raw(x1.data_ptr(), C1, x2.data_ptr() if C2 else None, C2, w.data_ptr(), layout, None, None,
                                     0, None, 1.0, y_b.data_ptr(), *dims, fws.data_ptr(), None, fws.data_ptr(), fb, None)
      assert rc != 0
  if blocks[1] is not None:
    call(lib, 'conv2d_dgrad_wp_f32', dy, w, layout, dx_b[0], C1, 0.0, dx_b[1], C2, 0.0, 1.0, *dims, blocks[1][1], amax, fws, fb)
    assert torch.equal(dx_a[0], dx_b[0])
    if C2:
      assert torch.equal(dx_a[1], dx_b[1])
  # weight gradient with the maxima its forward / data-gradient calls left behind == with its own passes
  have = (1 if blocks[0] is not None else 0) | (2 if blocks[1] is not None else 0)
  nbw = int(lib.conv2d_wgrad_ws_bytes(C1, C2, N, Cout, OH, OW, K, K))
  wws = torch.full((nbw // 4 + 64,), float('nan'), device=d)
  dw_a, dw_b = torch.zeros_like(w), torch.zeros_like(w)
  call(lib, 'conv2d_wgrad_f32', x1, C1, x2, C2, dy, dw_a, layout, 1.0, wws, wws.numel() * 4, *dims)
  call(lib, 'conv2d_wgrad_amax_f32', x1, C1, x2, C2, dy, dw_b, layout, 1.0, wws, wws.numel() * 4, *dims, amax, have)
  assert torch.equal(dw_a, dw_b)
  torch.cuda.synchronize()


def test_conv_adjoint_full_size(hip_lib):
  """Size-independent properties at the BASELINE size (DDPM++ 32x32, batch 128, 128->128 3x3):
  <conv(x), g> == <x, dgrad(g)> == <w, wgrad(x, g)>."""
  d = dev_of(hip_lib)
  N, C, H = 128, 128, 32
  x = rnd(N, C, H, H, seed=1).to(d)
  w = (rnd(C, C, 3, 3, seed=2) / 34.).to(d)
  g = rnd(N, C, H, H, seed=3).to(d)
  dims = (N, H, H, C, H, H, 3, 3, 1, 1)
  shape = (C, 0, N, H, H, C, 3, 3, 1, 1)
  fb = max(int(hip_lib.conv2d_fwd_ws_bytes(*shape)), int(hip_lib.conv2d_dgrad_ws_bytes(*shape)))
  assert fb > 0                                     # this is the shape the split kernel exists for
  fws = torch.zeros(fb // 4 + 64, device=d)
  y = torch.empty_like(x)
  call(hip_lib, 'conv2d_fwd_f32', x, C, None, 0, w, 0, None, None, 0, None, 1.0, y, *dims, fws, fb)
  dx = torch.empty_like(x)
  call(hip_lib, 'conv2d_dgrad_f32', g, w, 0, dx, C, 0.0, None, 0, 0.0, 1.0, *dims, fws, fb)
  ws = torch.zeros(int(hip_lib.conv2d_wgrad_ws_bytes(C, 0, N, C, H, H, 3, 3)) // 4 + 64, device=d)
  dw = torch.zeros_like(w)
  call(hip_lib, 'conv2d_wgrad_f32', x, C, None, 0, g, dw, 0, 1.0, ws, ws.numel() * 4, *dims)
  a = (y.double() * g.double()).sum().item()
  b = (x.double() * dx.double()).sum().item()
  c = (w.double() * dw.double()).sum().item()
  # The three inner products cancel heavily (16.7 M terms of either sign sum to a few hundred), so the tolerance is
  # set against the sum of |terms|: each term carries ~1e-6 relative error, random in sign -> ~1e-9 of that sum;
  # 1e-8 leaves a decade of margin and is still ~1e5 times smaller than what a dropped tap or tile would change.
  tol = 1e-8 * (y.double().abs() * g.double().abs()).sum().item()
  assert abs(a - b) <= tol and abs(a - c) <= tol, (a, b, c, tol)


RANGE_CASES = [
  # N, Cin, Cout, H, activation scale, weight scale, gradient scale, per-image spread (decades)
  (24, 128, 128, 32, 1.0, 0.03, 1.0e-4, 0.0),
  (43, 64, 128, 24, 300.0, 2.0e-4, 1.0, 0.0),        # large activations, small weights: the power-of-two scales do the work
  (32, 128, 128, 16, 1.0e-6, 5.0, 1.0e-9, 0.0),      # tiny magnitudes (fp16 would flush them without the scaling)
  (32, 128, 128, 16, 1.0, 0.03, 1.0e-3, 6.0),        # images whose magnitudes spread over six decades (loss weights)
  (24, 96, 160, 16, 2.0e4, 1.0e3, 1.0e5, 0.0),       # products far above fp16's largest value
]


@pytest.mark.parametrize('case', RANGE_CASES, ids=str)
def test_conv_split_dynamic_range(ref_lib, hip_lib, case):
  """The fp16 two-way split kernels (conv_x2.h) scale every operand tensor by a power of two taken from its |x| maximum:
  forward, data gradient and weight gradient must hold the usual 1e-4 of max|result| for magnitudes far outside fp16's
  range and for batches whose images differ by orders of magnitude."""
  N, Cin, Cout, H, xs, wsc, gs, spread = case
  ramp = (10.0 ** (-spread * torch.arange(N).float() / (N - 1))).view(N, 1, 1, 1) if spread else 1.0
  x = rnd(N, Cin, H, H, seed=1) * xs * ramp
  w = rnd(Cout, Cin, 3, 3, seed=2) * wsc
  dy = rnd(N, Cout, H, H, seed=3) * gs * ramp
  dims = (N, H, H, Cout, H, H, 3, 3, 1, 1)
  shape = (Cin, 0, N, H, H, Cout, 3, 3, 1, 1)

  def fn(lib, to):
    fb = max(int(lib.conv2d_fwd_ws_bytes(*shape)), int(lib.conv2d_dgrad_ws_bytes(*shape)))
    if lib.is_device:
      assert fb > 0 and int(lib.conv2d_variant(0, Cin, 0, N, H, H, Cout, H, H, 3, 3, 1, 1, 0)) == 5
    fws = to(torch.full((fb // 4 + 64,), float('nan'))) if fb else None
    xx, ww, dd = to(x), to(w), to(dy)
    y = to(torch.full((N, Cout, H, H), float('nan')))
    call(lib, 'conv2d_fwd_f32', xx, Cin, None, 0, ww, 0, None, None, 0, None, 1.0, y, *dims, fws, fb)
    dx = to(torch.full((N, Cin, H, H), float('nan')))
    call(lib, 'conv2d_dgrad_f32', dd, ww, 0, dx, Cin, 0.0, None, 0, 0.0, 1.0, *dims, fws, fb)
    nb = int(lib.conv2d_wgrad_ws_bytes(Cin, 0, N, Cout, H, H, 3, 3))
    ws = to(torch.full((max(nb // 4, 64) + 64,), float('nan')))
    dw = to(torch.zeros(Cout, Cin, 3, 3))
    call(lib, 'conv2d_wgrad_f32', xx, Cin, None, 0, dd, dw, 0, 1.0, ws, ws.numel() * 4, *dims)
    return {'y': y, 'dx': dx, 'dw': dw}

  outs = both(ref_lib, hip_lib, fn)
  compare(outs, 1e-4, 'conv split range')
  if spread:
    # per IMAGE: max-relative over the whole tensor would let an image at 1e-6 of the batch maximum be 100 % wrong
    for k in ('y', 'dx'):
      worst = per_image_error(outs[1][k], outs[0][k]).max().item()
      assert worst <= 2e-5, f'conv split range:{k}: worst per-image error {worst:.3e} (images spread over {spread} decades)'


def per_image_error(got, ref):
  """[N] max |got - ref| over each image, relative to that image's max |ref|."""
  g, r = got.detach().cpu().double().flatten(1), ref.detach().cpu().double().flatten(1)
  return (g - r).abs().max(1).values / r.abs().max(1).values.clamp_min(1e-300)


def test_conv_split_per_image_accuracy(ref_lib, hip_lib):
  """Where the split kernels stop being fp32 PER SAMPLE.  One power-of-two scale per operand TENSOR puts the batch maximum at
  [2^13, 2^14); an image whose magnitude is rho times the maximum keeps its fp16 hi term (11 bits) and a lo term that goes
  subnormal (absolute precision 2^-24 in scaled units) once rho < ~2^-16: its own relative error then grows like 1 / rho.
  Measured on MI355X (forward; the data gradient is the same): 2e-7 down to 1e-4 of the maximum, 9e-7 at 1e-5 (the 4-5
  decades the per-sample loss weights g^2 / sigma spread dy over in the VE configs), 8e-6 at 1e-6, 8e-5 at 1e-7, 9e-4 at
  1e-8, 9e-3 at 1e-9 -- i.e. ~ 2^-36.8 / rho.  Asserted per image for forward and data gradient over NINE decades:
  error <= max(3e-6, 8 * 2^-38.5 / rho), and fp32-level (1.5e-6) for every image within five decades of the maximum."""
  N, C, H, decades = 28, 128, 16, 9.0
  rho = 10.0 ** (-decades * torch.arange(N).double() / (N - 1))
  ramp = rho.float().view(N, 1, 1, 1)
  x = rnd(N, C, H, H, seed=1) * ramp
  w = rnd(C, C, 3, 3, seed=2) * 0.03
  dy = rnd(N, C, H, H, seed=3) * 1e-3 * ramp
  dims = (N, H, H, C, H, H, 3, 3, 1, 1)
  shape = (C, 0, N, H, H, C, 3, 3, 1, 1)

  def fn(lib, to):
    fb = max(int(lib.conv2d_fwd_ws_bytes(*shape)), int(lib.conv2d_dgrad_ws_bytes(*shape)))
    fws = to(torch.full((fb // 4 + 64,), float('nan'))) if fb else None
    xx, ww, dd = to(x), to(w), to(dy)
    y = to(torch.full((N, C, H, H), float('nan')))
    call(lib, 'conv2d_fwd_f32', xx, C, None, 0, ww, 0, None, None, 0, None, 1.0, y, *dims, fws, fb)
    dx = to(torch.full((N, C, H, H), float('nan')))
    call(lib, 'conv2d_dgrad_f32', dd, ww, 0, dx, C, 0.0, None, 0, 0.0, 1.0, *dims, fws, fb)
    return {'y': y, 'dx': dx}

  ref, got = both(ref_lib, hip_lib, fn)
  bound = torch.maximum(torch.tensor(3e-6, dtype=torch.float64), 8 * 2.0 ** -38.5 / rho)
  for k in ('y', 'dx'):
    e = per_image_error(got[k], ref[k])
    print(k, 'per-image error by decade:', [f'{rho[i].item():.0e}: {e[i].item():.1e}' for i in range(0, N, 3)])
    assert (e <= bound).all(), f'{k}: per-image errors {e.tolist()} exceed {bound.tolist()}'
    assert e[rho >= 0.99e-5].max().item() <= 1.5e-6


THIN_FULL = [
  # N, Cin, Cout, H, K  -- the thin-side layers at the BASELINE sizes
  (128, 3, 128, 32, 3),      # DDPM++ CIFAR-10 stem
  (128, 128, 3, 32, 3),      # ... and head
  (4, 3, 128, 256, 1),       # NCSN++ 256x256 input_skip Combine (3 -> C, 1x1)
  (4, 128, 3, 256, 3),       # ... output_skip conv (C -> 3)
  (4, 256, 3, 32, 3),        # output_skip at a deeper level (two weight blocks of 128 channels)
]


@pytest.mark.parametrize('case', THIN_FULL, ids=str)
def test_thin_conv_adjoint_full_size(hip_lib, case):
  """<conv(x), g> == <x, dgrad(g)> == <w, wgrad(x, g)> for the streaming kernels (conv_thin.h) at full size."""
  d = dev_of(hip_lib)
  N, Cin, Cout, H, K = case
  pad = K // 2
  x = rnd(N, Cin, H, H, seed=1).to(d)
  w = (rnd(Cout, Cin, K, K, seed=2) / float(np.sqrt(Cin * K * K))).to(d)
  g = rnd(N, Cout, H, H, seed=3).to(d)
  dims = (N, H, H, Cout, H, H, K, K, 1, pad)
  assert int(hip_lib.conv2d_variant(0, Cin, 0, N, H, H, Cout, H, H, K, K, 1, pad, 0)) == 4
  assert int(hip_lib.conv2d_variant(2, Cin, 0, N, H, H, Cout, H, H, K, K, 1, pad, 0)) == 4
  y = torch.full((N, Cout, H, H), float('nan'), device=d)
  call(hip_lib, 'conv2d_fwd_f32', x, Cin, None, 0, w, 0, None, None, 0, None, 1.0, y, *dims, None, 0)
  dx = torch.full((N, Cin, H, H), float('nan'), device=d)
  call(hip_lib, 'conv2d_dgrad_f32', g, w, 0, dx, Cin, 0.0, None, 0, 0.0, 1.0, *dims, None, 0)
  ws = torch.full((int(hip_lib.conv2d_wgrad_ws_bytes(Cin, 0, N, Cout, H, H, K, K)) // 4 + 64,), float('nan'), device=d)
  dw = torch.zeros_like(w)
  call(hip_lib, 'conv2d_wgrad_f32', x, Cin, None, 0, g, dw, 0, 1.0, ws, ws.numel() * 4, *dims)
  a = (y.double() * g.double()).sum().item()
  b = (x.double() * dx.double()).sum().item()
  c = (w.double() * dw.double()).sum().item()
  tol = 1e-8 * (y.double().abs() * g.double().abs()).sum().item()      # see test_conv_adjoint_full_size
  assert abs(a - b) <= tol and abs(a - c) <= tol, (a, b, c, tol)


# ---------------------------------------------------------------------------------------------------
# batched strided GEMM, softmax
# ---------------------------------------------------------------------------------------------------
GEMM_CASES = [
  # M, N, K, batch, a_kcontig, b_kcontig, bias_mode, beta
  (256, 256, 256, 6, False, False, 0, 0.0),     # S = Q^T K
  (256, 256, 256, 6, True, True, 0, 0.5),       # O = V P^T
  (64, 64, 256, 3, True, False, 0, 0.0),
  (512, 7, 128, 1, True, True, 2, 0.0),         # Linear: [B=7] outputs, bias along n... (here along n)
  (8, 512, 512, 1, True, True, 2, 0.0),         # temb MLP shape (batch 8)
  (130, 70, 33, 2, False, True, 1, 1.0),        # ragged
  (16, 16, 16, 5, False, False, 0, 0.0),        # mid-block attention at 4x4
]


@pytest.mark.parametrize('case', GEMM_CASES, ids=str)
def test_gemm(ref_lib, hip_lib, case):
  M, N, K, batch, akc, bkc, bias_mode, beta = case
  A = rnd(batch, M, K, seed=1) if akc else rnd(batch, K, M, seed=1)
  B = rnd(batch, N, K, seed=2) if bkc else rnd(batch, K, N, seed=2)
  C0 = rnd(batch, M, N, seed=3)
  bias = rnd(M if bias_mode == 1 else N, seed=4) if bias_mode else None
  sam, sak = (K, 1) if akc else (1, M)
  sbk, sbn = (1, K) if bkc else (N, 1)

  def fn(lib, to):
    c = to(C0.clone())
    call(lib, 'gemm_f32', to(A), sam, sak, M * K, to(B), sbk, sbn, N * K, c, N, 1, M * N, to(bias), bias_mode,
         M, N, K, batch, 0.75, beta)
    ct = to(C0.transpose(1, 2).contiguous().clone())     # transposed output (scm = 1)
    call(lib, 'gemm_f32', to(A), sam, sak, M * K, to(B), sbk, sbn, N * K, ct, 1, M, M * N, to(bias), bias_mode,
         M, N, K, batch, 0.75, beta)
    return {'c': c, 'ct': ct}

  compare(both(ref_lib, hip_lib, fn), 1e-4, 'gemm')


@pytest.mark.parametrize('rows,cols', [(64, 256), (37, 16), (10, 64), (5, 300), (1000, 1)])
def test_softmax(ref_lib, hip_lib, rows, cols):
  x, dy = rnd(rows, cols, seed=1) * 3, rnd(rows, cols, seed=2)

  def fn(lib, to):
    y = to(torch.zeros(rows, cols))
    call(lib, 'softmax_fwd_f32', to(x), y, rows, cols, 0.0625)
    dx = to(dy.clone())
    call(lib, 'softmax_bwd_f32', y, dx, dx, rows, cols, 0.0625)      # in place, as the attention op uses it
    return {'y': y, 'dx': dx}

  compare(both(ref_lib, hip_lib, fn), 1e-5, 'softmax')


# fused attention core (csrc/attention.hip) against the oracle's double-precision restatement of
# models/layerspp.py:95-99 and of its autograd.  Shapes: the 16x16 blocks of every shipped config (C = 256, T = 256),
# the mid-block of CelebA-64 (T = 64) and of CIFAR-10 / CelebA-HQ (T = 16), the `wide` test family (C = 192 / 96),
# a T that is neither a tile nor a block multiple, and the smallest C.
ATTN_CASES = [
  # (B, C, T, beta, magnitudes of q, k, v, do)
  (3, 256, 256, 0.0, (1., 1., 1., 1.)),
  (2, 256, 64, 0.0, (1., 1., 1., 1.)),
  (3, 256, 16, 1.0, (1., 1., 1., 1.)),
  (2, 192, 256, 0.5, (1., 1., 1., 1.)),
  (2, 96, 64, 0.0, (1., 1., 1., 1.)),
  (2, 64, 200, 0.0, (1., 1., 1., 1.)),
  (2, 32, 132, 1.0, (1., 1., 1., 1.)),
  (2, 128, 256, 0.0, (3e3, 2e-3, 5e-4, 1e-6)),        # operands far apart in magnitude: every tensor has its own scale
  (2, 256, 256, 0.0, (0.02, 30., 1e4, 1e3)),          # peaked softmax rows
]


@pytest.mark.parametrize('stacked', [False, True], ids=['separate', 'stacked'])
@pytest.mark.parametrize('case', ATTN_CASES, ids=lambda c: f'B{c[0]}C{c[1]}T{c[2]}b{c[3]}m{c[4][0]:g}')
def test_attention(ref_lib, hip_lib, case, stacked):
  """stacked: q, k, v (and dq, dk, dv) are the channel slices of one [B, 3C, T] tensor, addressed through the batch
  stride, as the engine's stacked q / k / v projection hands them over."""
  B, C, T, beta, (mq, mk, mv, mdo) = case
  assert hip_lib.attention_ok(B, C, T) == 1
  q, k, v, do = rnd(B, C, T, seed=1) * mq, rnd(B, C, T, seed=2) * mk, rnd(B, C, T, seed=3) * mv, rnd(B, C, T, seed=4) * mdo
  g0 = [rnd(B, C, T, seed=5 + i) * m * 0.1 for i, m in enumerate((mk, mq, mdo))]       # accumulated into when beta != 0
  scale = float(C) ** -0.5
  bs = 3 * C * T if stacked else C * T

  def fn(lib, to):
    o, lse, rec, delta = to(torch.zeros(B, C, T)), to(torch.zeros(B, T)), to(torch.zeros(1024)), to(torch.zeros(B, T))
    if stacked:
      qkv, g = to(torch.cat([q, k, v], 1)), to(torch.cat(g0, 1))
      ins = [qkv[0, i * C:].data_ptr() for i in range(3)]
      outs_ = [g[0, i * C:].data_ptr() for i in range(3)]
    else:
      keep = [to(q), to(k), to(v)] + [to(t.clone()) for t in g0]
      ins, outs_ = [t.data_ptr() for t in keep[:3]], [t.data_ptr() for t in keep[3:]]
    stream = torch.cuda.current_stream().cuda_stream if lib.is_device else 0
    lib.attention_fwd_f32(ins[0], ins[1], ins[2], bs, o.data_ptr(), lse.data_ptr(), rec.data_ptr(), B, C, T, scale, stream)
    lib.attention_bwd_f32(ins[0], ins[1], ins[2], bs, to(do).data_ptr(), lse.data_ptr(), rec.data_ptr(), delta.data_ptr(),
                          outs_[0], beta, outs_[1], beta, outs_[2], beta, bs, B, C, T, scale, stream)
    if lib.is_device:
      torch.cuda.synchronize()
    if stacked:
      dq, dk, dv = (g[:, i * C:(i + 1) * C].contiguous() for i in range(3))
    else:
      dq, dk, dv = keep[3:]
    return {'o': o, 'lse': lse, 'delta': delta, 'dq': dq, 'dk': dk, 'dv': dv}

  outs = both(ref_lib, hip_lib, fn)
  ref, got = outs
  for name in ('o', 'dq', 'dk', 'dv'):
    close(got[name], ref[name], rtol=1e-4, what=f'attention:{name}')
  close(got['lse'], ref['lse'], rtol=1e-5, atol=1e-5, what='attention:lse')
  close(got['delta'], ref['delta'], rtol=1e-4, what='attention:delta')


def test_attention_unsupported_shapes_are_refused(hip_lib):
  """Shapes outside the fused kernels' range return STK_EUNSUPPORTED (the engine plans them as GEMMs + softmax)."""
  assert hip_lib.attention_ok(2, 48, 64) == 0 and hip_lib.attention_ok(2, 512, 64) == 0
  assert hip_lib.attention_ok(2, 64, 1024) == 0 and hip_lib.attention_ok(2, 64, 30) == 0
  x = torch.zeros(2 * 48 * 64, device='cuda')
  r = torch.zeros(1024, device='cuda')
  rc = hip_lib.attention_fwd_f32.raw(x.data_ptr(), x.data_ptr(), x.data_ptr(), 48 * 64, x.data_ptr(), r.data_ptr(), r.data_ptr(),
                                     2, 48, 64, 0.1, 0)
  assert rc != 0


# ---------------------------------------------------------------------------------------------------
# optimizer side
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('n', [1, 1000, 1 << 20, (1 << 22) + 3])
def test_optimizer_kernels(ref_lib, hip_lib, n):
  p, g, m = rnd(n, seed=1), rnd(n, seed=2) * 0.01, rnd(n, seed=3) * 0.01
  v = (rnd(n, seed=4) * 0.01) ** 2
  sh = rnd(n, seed=5)

  def fn(lib, to):
    o = {}
    ws, ss = to(torch.zeros(2048)), to(torch.zeros(1))
    gg = to(g.clone())
    call(lib, 'sumsq_f32', gg, n, ss, ws)
    o['sumsq'] = ss
    pp, mm, vv = to(p.clone()), to(m.clone()), to(v.clone())
    call(lib, 'adam_f32', pp, gg, mm, vv, n, 2e-4, 0.9, 0.999, 1e-8, 0.0, 0, 1 - 0.9 ** 3, 1 - 0.999 ** 3, ss, 1.0)
    o.update(p=pp, g=gg, m=mm, v=vv)
    p2, g2, m2, v2 = to(p.clone()), to(g.clone()), to(m.clone()), to(v.clone())
    call(lib, 'adam_f32', p2, g2, m2, v2, n, 1e-3, 0.9, 0.99, 1e-8, 0.01, 1, 0.1, 0.01, None, -1.0)
    o.update(p_w=p2, m_w=m2, v_w=v2)
    s = to(sh.clone())
    call(lib, 'ema_f32', s, pp, n, 1 - 0.9999)
    o['ema'] = s
    return o

  compare(both(ref_lib, hip_lib, fn), 2e-5, 'optimizer')


# ---------------------------------------------------------------------------------------------------
# sample post-processing (sampling_lib.py:36-57)
# ---------------------------------------------------------------------------------------------------
def test_samples_to_uint8_and_get_samples(st, ref_lib, hip_lib, tmp_path):
  x = rnd(37, 3, 32, 32, seed=5) * 0.4 + 0.5
  want = np.clip(x.permute(0, 2, 3, 1).numpy() * 255., 0, 255).astype(np.uint8)
  assert np.array_equal(st.sampling_lib.samples_to_uint8(x, backend=ref_lib), want)
  assert np.array_equal(st.sampling_lib.samples_to_uint8(x.to(dev_of(hip_lib))), want)       # bit-exact on the device

  cfg = st.configs.get_config('cifar10_ddpmpp_nll_st')
  calls = []

  def sampling_fn(model):
    calls.append(1)
    return x[:16].to(dev_of(hip_lib)), 7

  out = st.sampling_lib.get_samples(cfg, None, None, sampling_fn, step=3, r=0, sample_dir=str(tmp_path))
  assert np.array_equal(out, want[:16])
  d = st.sampling_lib.get_dir_name(cfg, str(tmp_path), 3)
  assert np.array_equal(np.load(d + '/samples_0.npz')['samples'], want[:16])
  again = st.sampling_lib.get_samples(cfg, None, None, sampling_fn, step=3, r=0, sample_dir=str(tmp_path))
  assert len(calls) == 1 and np.array_equal(again, out)          # second call reads the file, as the reference does


def test_preprocess_u8(st, ref_lib, hip_lib):
  """Device input-pipeline tail: bit-identical to the checker (shared counter RNG), all four flag combinations."""
  g = torch.Generator().manual_seed(2)
  img = torch.randint(0, 256, (33, 16, 12, 3), generator=g, dtype=torch.uint8)
  cfg = st.configs.get_config('cifar10_ddpmpp_nll_st')
  for dequant, flip, centered in ((False, False, True), (True, True, False), (True, False, True), (False, True, False)):
    cfg.data.dequantization = 'uniform' if dequant else 'none'
    cfg.data.random_flip, cfg.data.centered = flip, centered
    a = st.datasets.device_batch(cfg, img, seed=1234, backend=ref_lib)
    b = st.datasets.device_batch(cfg, img.to(dev_of(hip_lib)), seed=1234)
    assert torch.equal(a, b.cpu())


def test_bias_grad_long_rows(ref_lib, hip_lib):
  """64x64 maps: rows of 4096 floats take the one-workgroup-per-row kernel."""
  N, C, HW, TS = 3, 10, 4096, 16
  dy = rnd(N, C, 64, 64, seed=3)
  db0 = rnd(C, seed=4)

  def fn(lib, to):
    d = to(dy)
    ws = to(torch.zeros(N * C))
    dt = to(torch.zeros(N, TS)); db = to(db0.clone())
    call(lib, 'bias_grad_f32', d, N, C, HW, 0.5, dt.data_ptr() + 4 * 4, TS, db, ws)
    db2 = to(db0.clone())
    call(lib, 'bias_grad_f32', d, N, C, HW, 1.0, None, 0, db2, ws)
    return {'dtemb': dt, 'dbias': db, 'dbias_only': db2}

  compare(both(ref_lib, hip_lib, fn), 1e-5, 'bias_grad')
