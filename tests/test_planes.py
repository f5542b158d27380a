"""Planes (include/stk.h "Planes", csrc/conv_pl.h): the oracle's restatement of the format against numpy's IEEE
binary16 conversion (CPU), and the HIP kernels against the oracle (GPU): the split pass BIT-exact (it is byte work:
scale by a power of two, two round-to-nearest conversions, a fixed layout), the plane-consuming convolutions at the
tolerance of the other convolution tests."""
import ctypes

import numpy as np
import pytest
import torch

from _util import call, dev_of, rnd


def _planes_np(x, amax):
  """numpy model of the format: [split][n][c / 32][pixel][c % 32] float16 of s x."""
  N, C, HW = x.shape
  m = float(np.max(amax))
  s = 1.0 if m == 0 else 2.0 ** (13 - int(np.floor(np.log2(m))))
  Cb = (C + 31) // 32
  xp = np.zeros((N, Cb * 32, HW), np.float32)
  xp[:, :C] = x * np.float32(s)
  hi = xp.astype(np.float16)
  lo = (xp - hi.astype(np.float32)).astype(np.float16)
  lay = lambda a: a.reshape(N, Cb, 32, HW).transpose(0, 1, 3, 2)
  return np.stack([lay(hi), lay(lo)]), s


def _split(lib, x, amax):
  N, C, HW = x.shape
  d = dev_of(lib)
  nb = int(lib.planes_bytes(N, C, HW))
  out = torch.full((nb,), 0xAA, dtype=torch.uint8, device=d)
  call(lib, 'split_planes_f32', x.to(d), N, C, HW, amax.to(d), amax.numel(), out)
  return out.cpu().numpy().view(np.float16).reshape(2, N, (C + 31) // 32, HW, 32)


SPLIT_CASES = [(2, 32, 64), (3, 96, 256), (1, 40, 20), (2, 128, 1024), (5, 3, 7)]


@pytest.mark.parametrize('case', SPLIT_CASES, ids=str)
def test_oracle_planes_match_numpy_float16(ref_lib, case):
  N, C, HW = case
  x = rnd(N, C, HW, seed=N + C) * 3.0
  x[0, 0, 0] = 0.0
  x[-1, -1, -1] = 1e-9          # far below the maximum: its second term lives in fp16's subnormal range
  amax = torch.zeros(256)
  amax[7] = x.abs().max()
  got = _split(ref_lib, x, amax)
  want, s = _planes_np(x.numpy(), amax.numpy())
  assert np.array_equal(got.view(np.uint16), want.view(np.uint16))
  # hi + lo reproduces s x to 2^-22 relative (or 2^-25 absolute in scaled units)
  rec = (got[0].astype(np.float64) + got[1].astype(np.float64))
  ref = want[0].astype(np.float64) * 0 + (_planes_np(x.numpy(), amax.numpy())[0][0].astype(np.float64) * 0)
  xs = np.zeros((N, (C + 31) // 32 * 32, HW)); xs[:, :C] = x.numpy().astype(np.float64) * s
  xs = xs.reshape(N, -1, 32, HW).transpose(0, 1, 3, 2)
  assert np.all(np.abs(rec - xs) <= np.maximum(np.abs(xs) * 2.0 ** -22, 2.0 ** -25))


def test_oracle_amax_record_and_apriori_bound(ref_lib):
  """Any record with the same maximum gives the same planes; a larger (a-priori) bound only moves the scale."""
  x = rnd(2, 64, 48, seed=3)
  a = torch.zeros(256); a[0] = x.abs().max()
  b = torch.zeros(256); b[200] = x.abs().max(); b[3] = 0.5 * x.abs().max()
  assert np.array_equal(_split(ref_lib, x, a).view(np.uint16), _split(ref_lib, x, b).view(np.uint16))
  c = torch.zeros(256); c[0] = 16 * x.abs().max()
  p16 = _split(ref_lib, x, c)
  assert np.array_equal((p16[0].astype(np.float32) * 16).astype(np.float16).view(np.uint16),
                        _split(ref_lib, x, a)[0].view(np.uint16))
  part = torch.full((256,), -1.0)
  call(ref_lib, 'amax_partial_f32', x, x.numel(), part)
  assert float(part.max()) == float(x.abs().max()) and float(part.min()) >= 0


# ---- GPU ------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize('case', SPLIT_CASES + [(128, 128, 1024), (16, 256, 64)], ids=str)
def test_split_planes_bit_exact(ref_lib, hip_lib, case):
  N, C, HW = case
  x = rnd(N, C, HW, seed=N + C) * 3.0
  x[0, 0, 0] = 0.0
  x[-1, -1, -1] = 1e-9
  d = dev_of(hip_lib)
  part = torch.full((256,), float('nan'), device=d)
  call(hip_lib, 'amax_partial_f32', x.to(d), x.numel(), part)
  part = part.cpu()
  assert float(part.max()) == float(x.abs().max())
  if N * C * HW > 4e6:       # the oracle's triple loop is slow: check a random subset of images
    idx = [0, N // 2, N - 1]
    got = _split(hip_lib, x, part)[:, idx]
    want = _split(ref_lib, x[idx].contiguous(), part)
  else:
    got, want = _split(hip_lib, x, part), _split(ref_lib, x, part)
  assert np.array_equal(got.view(np.uint16), want.view(np.uint16))


PL_CONV_CASES = [
  # N, C, H, W, Cout, K, layout, temb, res, div
  (8, 128, 32, 32, 128, 3, 0, 1, 1, 1),
  (4, 96, 16, 16, 96, 3, 0, 0, 0, 0),
  (6, 256, 8, 8, 256, 3, 0, 1, 0, 1),       # K-split (few tiles)
  (3, 256, 4, 4, 256, 3, 0, 0, 1, 0),
  (96, 256, 16, 16, 256, 1, 0, 0, 1, 1),    # 1x1 Conv2d (needs >= 192 tiles: a 1x1 layer has too few chunks to split K)
  (96, 256, 16, 16, 256, 1, 1, 0, 0, 0),    # NIN
  (5, 128, 12, 20, 160, 3, 0, 0, 0, 0),     # ragged map, 160 rows
]
# small problems (few 128 x 128 tiles: K split + slab sum) of the shapes of the 8x8 / 4x4 levels and of the batch-4 regime
PL_CONV_CASES = PL_CONV_CASES + [
  (4, 256, 32, 32, 256, 3, 0, 1, 0, 1),     # 32x32 at batch 4
  (37, 256, 4, 4, 192, 3, 0, 1, 1, 1),      # 4x4: a ragged last tile (592 pixels), 192 rows
  (2, 96, 8, 8, 96, 3, 0, 1, 1, 1),         # the 'wide' test family's channel count (96 rows)
]


@pytest.mark.gpu
@pytest.mark.parametrize('case', PL_CONV_CASES, ids=str)
def test_conv_from_planes(ref_lib, hip_lib, case):
  """Forward and data gradient with the activation operand given as planes, HIP vs the oracle (which decodes the
  planes and runs its double-accumulating convolution), plus agreement with the HIP fp32-input call."""
  N, C, H, W, Cout, K, layout, use_temb, use_res, use_div = case
  x = rnd(N, C, H, W, seed=1)
  w = (rnd(Cout, C, K, K, seed=3) if layout == 0 else rnd(C, Cout, seed=3)) * (1.0 / np.sqrt(C * K * K))
  bias = rnd(Cout, seed=4)
  temb = rnd(N, Cout, seed=5) if use_temb else None
  res = rnd(N, Cout, H, W, seed=6) if use_res else None
  div = float(np.float32(np.sqrt(2.))) if use_div else 1.0
  dy = rnd(N, Cout, H, W, seed=7)
  g1 = rnd(N, C, H, W, seed=8)
  assert int(hip_lib.conv2d_pl_ok(0, C, 0, N, H, W, Cout, K, K, 1, K // 2)) == 1
  assert int(ref_lib.conv2d_pl_ok(0, C, 0, N, H, W, Cout, K, K, 1, K // 2)) == 1
  assert int(hip_lib.conv2d_pl_ok(1, C, 0, N, H, W, Cout, K, K, 1, K // 2)) == 1

  def run(lib):
    d = dev_of(lib)
    to = lambda t: None if t is None else t.to(d)
    shape = (C, 0, N, H, W, Cout, K, K, 1, K // 2)
    fb = max(int(lib.conv2d_fwd_ws_bytes(*shape)), int(lib.conv2d_dgrad_ws_bytes(*shape)), 256)
    ws = torch.zeros(fb // 4 + 64, device=d)
    ax, ay = torch.zeros(256, device=d), torch.zeros(256, device=d)
    xd, dyd = to(x), to(dy)
    call(lib, 'amax_partial_f32', xd, xd.numel(), ax)
    call(lib, 'amax_partial_f32', dyd, dyd.numel(), ay)
    xp = torch.zeros(int(lib.planes_bytes(N, C, H * W)), dtype=torch.uint8, device=d)
    yp = torch.zeros(int(lib.planes_bytes(N, Cout, H * W)), dtype=torch.uint8, device=d)
    call(lib, 'split_planes_f32', xd, N, C, H * W, ax, 256, xp)
    call(lib, 'split_planes_f32', dyd, N, Cout, H * W, ay, 256, yp)
    y = torch.zeros(N, Cout, H, W, device=d)
    call(lib, 'conv2d_fwd_pl_f32', xp, ax, C, to(w), layout, to(bias), to(temb), Cout if use_temb else 0, to(res), div, y,
         N, H, W, Cout, K, K, None, ws, fb)
    dx = to(g1.clone())
    call(lib, 'conv2d_dgrad_pl_f32', yp, ay, to(w), layout, dx, C, 1.0, None, 0, 0.0, 0.5, N, H, W, Cout, K, K, None, ws, fb)
    # the fp32-input calls of the same library
    y32 = torch.zeros(N, Cout, H, W, device=d)
    call(lib, 'conv2d_fwd_f32', xd, C, None, 0, to(w), layout, to(bias), to(temb), Cout if use_temb else 0, to(res), div,
         y32, N, H, W, Cout, H, W, K, K, 1, K // 2, ws, fb)
    dx32 = to(g1.clone())
    call(lib, 'conv2d_dgrad_f32', dyd, to(w), layout, dx32, C, 1.0, None, 0, 0.0, 0.5, N, H, W, Cout, H, W, K, K, 1, K // 2,
         ws, fb)
    return {k: v.cpu() for k, v in dict(y=y, dx=dx, y32=y32, dx32=dx32).items()}

  r, h = run(ref_lib), run(hip_lib)
  for k in ('y', 'dx'):
    scale = r[k].abs().max().item()
    assert (h[k] - r[k]).abs().max().item() <= 1e-4 * scale, k
    assert (h[k] - h[k + '32']).abs().max().item() <= 2e-5 * scale, k + ' vs the fp32-input call'


@pytest.mark.gpu
@pytest.mark.parametrize('case', [(16, 128, 128, 8, 8, 256), (5, 256, 128, 4, 4, 128), (4, 128, 256, 16, 16, 256)], ids=str)
def test_small_map_data_gradient_into_two_sources(ref_lib, hip_lib, case):
  """The K-split data gradient of a small map with the rows routed to the two sources of a concatenation (the first convolution
  of an up-path block, ncsnpp.py:368): dx1 overwritten, dx2 accumulated, against the oracle."""
  N, C1, C2, H, W, Cout = case
  dy = rnd(N, Cout, H, W, seed=7)
  w = rnd(Cout, C1 + C2, 3, 3, seed=3) / np.sqrt((C1 + C2) * 9.)
  g2 = rnd(N, C2, H, W, seed=9)
  out = {}
  for name, lib in (('ref', ref_lib), ('hip', hip_lib)):
    d = dev_of(lib)
    shape = (C1, C2, N, H, W, Cout, 3, 3, 1, 1)
    fb = max(int(lib.conv2d_dgrad_ws_bytes(*shape)), 256)
    ws = torch.zeros(fb // 4 + 64, device=d)
    ay = torch.zeros(256, device=d)
    dyd = dy.to(d)
    call(lib, 'amax_partial_f32', dyd, dyd.numel(), ay)
    yp = torch.zeros(int(lib.planes_bytes(N, Cout, H * W)), dtype=torch.uint8, device=d)
    call(lib, 'split_planes_f32', dyd, N, Cout, H * W, ay, 256, yp)
    dx1 = torch.full((N, C1, H, W), float('nan'), device=d)
    dx2 = g2.clone().to(d)
    call(lib, 'conv2d_dgrad_pl_f32', yp, ay, w.to(d), 0, dx1, C1, 0.0, dx2, C2, 1.0, 0.7, N, H, W, Cout, 3, 3, None, ws, fb)
    out[name] = (dx1.cpu(), dx2.cpu())
  for a, b in zip(out['hip'], out['ref']):
    assert (a - b).abs().max().item() <= 1e-4 * b.abs().max().item()


@pytest.mark.gpu
def test_conv_from_planes_apriori_bound(ref_lib, hip_lib):
  """A scale record holding a loose a-priori bound (what GroupNorm writes): 64x the true maximum still gives fp32-level
  results, because the second split term absorbs what the first loses."""
  N, C, H, Cout = 4, 128, 16, 128
  x = rnd(N, C, H, H, seed=1)
  w = rnd(Cout, C, 3, 3, seed=3) / np.sqrt(C * 9.)
  ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
  d = dev_of(hip_lib)
  shape = (C, 0, N, H, H, Cout, 3, 3, 1, 1)
  fb = int(hip_lib.conv2d_fwd_ws_bytes(*shape))
  ws = torch.zeros(fb // 4 + 64, device=d)
  for slack in (1.0, 64.0, 4096.0):
    rec = torch.zeros(256); rec[0] = slack * x.abs().max()
    rec = rec.to(d)
    xp = torch.zeros(int(hip_lib.planes_bytes(N, C, H * H)), dtype=torch.uint8, device=d)
    call(hip_lib, 'split_planes_f32', x.to(d), N, C, H * H, rec, 256, xp)
    y = torch.zeros(N, Cout, H, H, device=d)
    call(hip_lib, 'conv2d_fwd_pl_f32', xp, rec, C, w.to(d), 0, None, None, 0, None, 1.0, y, N, H, H, Cout, 3, 3, None, ws, fb)
    err = (y.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    assert err <= 3e-6, (slack, err)


GN_PL_CASES = [
  # N, C1, C2, HW, G, act, drop
  (4, 128, 0, 1024, 32, 1, 0.0),      # fused: 4 channels per group, 4 passes
  (3, 256, 0, 256, 32, 1, 0.1),       # fused: 8 per group, dropout
  (5, 256, 0, 64, 32, 0, 0.0),        # fused, no activation (attention's GroupNorm)
  (6, 128, 128, 16, 32, 1, 0.0),      # fused, two sources, 4x4 maps
  (2, 256, 256, 256, 32, 1, 0.0),     # fused: 16 per group, two sources
  (2, 256, 128, 64, 32, 1, 0.0),      # 12 per group: the unfused route (bound + split after the fp32 kernel)
  (2, 96, 0, 256, 24, 1, 0.0),        # the 'wide' test family
]


@pytest.mark.gpu
@pytest.mark.parametrize('case', GN_PL_CASES, ids=str)
def test_gn_forward_to_planes(ref_lib, hip_lib, case):
  """stk_gn_fwd_pl_f32 on the HIP library against the oracle's composition (GroupNorm restatement, bound, split):
  the scale record bit for bit, mean / rstd / y at the GroupNorm tolerance, and the planes decoded back (hi + lo) / s
  against y -- they must carry y to 2^-22."""
  N, C1, C2, HW, G, act, drop = case
  C = C1 + C2
  x1 = rnd(N, C1, HW, seed=1) * 2 + 0.3
  x2 = rnd(N, C2, HW, seed=2) if C2 else None
  gamma, beta = rnd(C, seed=3) * 0.5 + 1.0, rnd(C, seed=4) * 0.2
  out = {}
  for name, lib in (('ref', ref_lib), ('hip', hip_lib)):
    d = dev_of(lib)
    to = lambda t: None if t is None else t.to(d)
    y = torch.zeros(N, C, HW, device=d)
    mean, rstd = torch.zeros(N * G, device=d), torch.zeros(N * G, device=d)
    rec = torch.full((256,), -1.0, device=d)
    pl = torch.full((int(lib.planes_bytes(N, C, HW)),), 0xAA, dtype=torch.uint8, device=d)
    ws = torch.zeros(int(lib.gn_ws_bytes(N, C, HW, G)) // 4 + 64, device=d)
    call(lib, 'gn_fwd_pl_f32', to(x1), C1, to(x2), C2, to(gamma), to(beta), y, pl, rec, mean, rstd, N, HW, G, 1e-6, act,
         drop, 1234, None, ws)
    out[name] = dict(y=y.cpu(), mean=mean.cpu(), rstd=rstd.cpu(), rec=rec.cpu(),
                     pl=pl.cpu().numpy().view(np.float16).reshape(2, N, (C + 31) // 32, HW, 32))
    if name == 'hip' and int(lib.gn_fwd_pl_fused(C1, C2, HW, G)):
      # y = NULL: the planes must not change
      pl2 = torch.full_like(pl, 0x55)
      call(lib, 'gn_fwd_pl_f32', to(x1), C1, to(x2), C2, to(gamma), to(beta), None, pl2, rec, mean, rstd, N, HW, G, 1e-6, act,
           drop, 1234, None, ws)
      assert torch.equal(pl2, pl)
  r, h = out['ref'], out['hip']
  assert torch.equal(r['rec'], h['rec']) and float(r['rec'][0]) >= float(r['y'].abs().max())
  for k in ('y', 'mean', 'rstd'):
    assert (h[k] - r[k]).abs().max().item() <= 1e-5 * max(r[k].abs().max().item(), 1.0), k
  m = float(h['rec'].max())
  s = 2.0 ** (13 - int(np.floor(np.log2(m))))
  dec = (h['pl'][0].astype(np.float64) + h['pl'][1].astype(np.float64)) / s          # [N, Cb, HW, 32]
  dec = dec.transpose(0, 1, 3, 2).reshape(N, -1, HW)[:, :C]
  yy = h['y'].double().numpy()
  assert np.abs(dec - yy).max() <= max(2.0 ** -22 * np.abs(yy).max(), 2.0 ** -25 / s)
  assert np.all(h['pl'].reshape(2, N, -1, HW, 32).transpose(0, 1, 2, 4, 3).reshape(2, N, -1, HW)[:, :, C:] == 0)


@pytest.mark.gpu
@pytest.mark.parametrize('case', [(4, 128, 128, 32, 128), (3, 256, 0, 16, 256), (6, 256, 256, 4, 256), (2, 128, 0, 8, 256)], ids=str)
def test_gn_forward_leaves_source_records_for_the_shortcut(ref_lib, hip_lib, case):
  """stk_gn_fwd_pl_max_f32: the planes of stk_gn_fwd_pl_f32 bit for bit, plus max |x1| / max |x2| in caller-zeroed
  records (exact); stk_conv2d_fwd_rec_f32 on those records == stk_conv2d_fwd_wp_f32 measuring them itself, bit for bit
  (the 1x1 shortcut convolution of a ResnetBlock reads the block input the GroupNorm has just read)."""
  N, C1, C2, H, Cout = case
  C, HW, G = C1 + C2, H * H, 32
  x1 = rnd(N, C1, H, H, seed=1) * torch.logspace(-2, 1, N)[:, None, None, None]
  x2 = rnd(N, C2, H, H, seed=2) * 3 if C2 else None
  gamma, beta = rnd(C, seed=3) * 0.5 + 1.0, rnd(C, seed=4) * 0.2
  w, bias = rnd(Cout, C, 1, 1, seed=5) * 0.05, rnd(Cout, seed=6)
  for lib in (ref_lib, hip_lib):
    d = dev_of(lib)
    to = lambda t: None if t is None else t.to(d)
    assert int(lib.gn_fwd_pl_fused(C1, C2, HW, G)) == 1
    mean, rstd = torch.zeros(N * G, device=d), torch.zeros(N * G, device=d)
    rec, rec2 = torch.zeros(256, device=d), torch.zeros(256, device=d)
    pl = torch.zeros(int(lib.planes_bytes(N, C, HW)), dtype=torch.uint8, device=d)
    pl2 = torch.zeros_like(pl)
    ws = torch.zeros(int(lib.gn_ws_bytes(N, C, HW, G)) // 4 + 64, device=d)
    amax = torch.zeros(768, device=d)
    call(lib, 'gn_fwd_pl_f32', to(x1), C1, to(x2), C2, to(gamma), to(beta), None, pl, rec, mean, rstd, N, HW, G, 1e-6, 1,
         0.0, 1, None, ws)
    call(lib, 'gn_fwd_pl_max_f32', to(x1), C1, to(x2), C2, to(gamma), to(beta), None, pl2, rec2, mean, rstd, N, HW, G, 1e-6, 1,
         0.0, 1, None, ws, amax, amax[256:] if C2 else None)
    assert torch.equal(pl, pl2) and torch.equal(rec, rec2)
    assert float(amax[:256].max()) == float(x1.abs().max())
    if C2:
      assert float(amax[256:512].max()) == float(x2.abs().max())
    assert float(amax[512:].abs().max()) == 0.0
    # the shortcut convolution on those records
    dims = (N, H, H, Cout, H, H, 1, 1, 1, 0)
    fb = int(lib.conv2d_fwd_ws_bytes(C1, C2, N, H, H, Cout, 1, 1, 1, 0))
    cws = torch.zeros(fb // 4 + 64, device=d)
    y_rec, y_own = torch.zeros(N, Cout, H, H, device=d), torch.zeros(N, Cout, H, H, device=d)
    own = torch.zeros(768, device=d)
    call(lib, 'conv2d_fwd_rec_f32', to(x1), C1, to(x2), C2, to(w), 0, to(bias), None, 0, None, 1.0, y_rec, *dims, None, amax, cws, fb)
    call(lib, 'conv2d_fwd_wp_f32', to(x1), C1, to(x2), C2, to(w), 0, to(bias), None, 0, None, 1.0, y_own, *dims, None, own, cws, fb)
    assert torch.equal(y_rec, y_own)


@pytest.mark.gpu
@pytest.mark.parametrize('case', [(8, 128, 1024), (5, 256, 64), (3, 96, 256), (2, 64, 4096)], ids=str)
def test_bias_grad_with_scale_record(ref_lib, hip_lib, case):
  """stk_bias_grad_amax_f32: the bias / time-embedding gradients of stk_bias_grad_f32 plus the per-channel |dy| maxima
  as a scale record (bit-exact: a maximum has no rounding)."""
  N, C, HW = case
  dy = rnd(N, C, HW, seed=5) * torch.logspace(-3, 1, N)[:, None, None]
  res = {}
  for name, lib in (('ref', ref_lib), ('hip', hip_lib)):
    d = dev_of(lib)
    db, dt = torch.ones(C, device=d), torch.zeros(N, C + 8, device=d)
    rec = torch.full((256,), float('nan'), device=d)
    ws = torch.zeros(N * C + 64, device=d)
    call(lib, 'bias_grad_amax_f32', dy.to(d), N, C, HW, 0.5, dt, C + 8, db, rec, ws)
    rec2 = torch.full((256,), float('nan'), device=d)
    call(lib, 'bias_grad_amax_f32', dy.to(d), N, C, HW, 0.5, None, 0, None, rec2, ws)       # record only
    res[name] = (db.cpu(), dt.cpu(), rec.cpu(), rec2.cpu())
  (rdb, rdt, rrec, _), (hdb, hdt, hrec, hrec2) = res['ref'], res['hip']
  assert float(hrec.max()) == float(dy.abs().max()) == float(rrec.max())
  assert not torch.isnan(hrec).any() and float(hrec.min()) >= 0
  assert float(hrec2.max()) == float(hrec.max())
  if HW < 4096:
    assert torch.equal(hrec, rrec)
  assert (hdb - rdb).abs().max().item() <= 1e-5 * rdb.abs().max().item()
  assert (hdt - rdt).abs().max().item() <= 1e-5 * rdt.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize('beta', [0.0, 1.0], ids=lambda b: f'beta{b:g}')
@pytest.mark.parametrize('case', [(8, 128, 1024), (5, 256, 64), (3, 96, 256), (2, 64, 4096), (3, 40, 25)], ids=str)
def test_bias_grad_with_scale_record_and_residual(ref_lib, hip_lib, case, beta):
  """stk_bias_grad_amax_res_f32 = stk_bias_grad_amax_f32 + dres = alpha dy + beta dres in the same pass over dy (the
  skip connection's gradient): the sums and the record as the plain call, dres to one rounding of the axpby it replaces."""
  N, C, HW = case
  dy = rnd(N, C, HW, seed=5) * torch.logspace(-3, 1, N)[:, None, None]
  r0 = rnd(N, C, HW, seed=6)
  res = {}
  for name, lib in (('ref', ref_lib), ('hip', hip_lib)):
    d = dev_of(lib)
    out = []
    for fused in (True, False):
      db, dt = torch.ones(C, device=d), torch.zeros(N, C + 8, device=d)
      rec = torch.full((256,), float('nan'), device=d)
      ws = torch.zeros(N * C + 64, device=d)
      dres = r0.to(d).clone() if beta else torch.full((N, C, HW), float('nan'), device=d)
      if fused:
        call(lib, 'bias_grad_amax_res_f32', dy.to(d), N, C, HW, 0.5, dt, C + 8, db, rec, dres, beta, ws)
      else:
        call(lib, 'bias_grad_amax_f32', dy.to(d), N, C, HW, 0.5, dt, C + 8, db, rec, ws)
        call(lib, 'axpby_f32', dy.to(d), 0.5, dres, beta, dres, N * C * HW)
      out.append((db.cpu(), dt.cpu(), rec.cpu(), dres.cpu()))
    res[name] = out
  (hf, hp), (rf, _) = res['hip'], res['ref']
  assert torch.equal(hf[2], hp[2]) and float(hf[2].max()) == float(dy.abs().max())       # the record: bit-exact
  assert torch.equal(hf[0], hp[0]) and torch.equal(hf[1], hp[1])                           # same summation order
  assert (hf[3] - hp[3]).abs().max().item() <= 1e-6 * hp[3].abs().max().item()
  assert (hf[3] - rf[3]).abs().max().item() <= 1e-6 * rf[3].abs().max().item()
  assert (hf[0] - rf[0]).abs().max().item() <= 1e-5 * rf[0].abs().max().item()


WGRAD_PL_CASES = [
  # N, Cin, Cout, H
  (8, 128, 128, 32),      # COLS = 32, one row per chunk
  (4, 64, 96, 16),        # COLS = 16, Cout not a multiple of 128 (dead waves)
  (8, 256, 160, 8),       # COLS = 8, two co tiles, the second one ragged
  (2, 32, 32, 64),        # W > 32: a chunk is half a row
  (3, 96, 128, 16),
  (16, 256, 256, 4),      # COLS = 4: a chunk is two whole images, each with its own halo tile
  (18, 64, 160, 4),       # ... ragged co tile, K split with a short last slab
  (1, 32, 64, 128),       # W = 128: four chunks per row, the dy halo columns are the neighbouring chunks' pixels (round 6)
]


@pytest.mark.gpu
@pytest.mark.parametrize('case', WGRAD_PL_CASES, ids=str)
def test_wgrad_from_planes(ref_lib, hip_lib, case):
  """3x3 weight gradient with x and dy as planes (LDS-DMA + transpose reads) against the oracle, accumulating into a
  non-zero dw with alpha != 1, and against the HIP fp32-operand weight gradient."""
  N, Cin, Cout, H = case
  x = rnd(N, Cin, H, H, seed=1)
  dy = rnd(N, Cout, H, H, seed=2) * torch.logspace(-2, 0, N)[:, None, None, None]
  dw0 = rnd(Cout, Cin, 3, 3, seed=3)
  assert int(hip_lib.conv2d_wgrad_pl_ok(N, H, H, Cin, Cout)) == 1 and int(ref_lib.conv2d_wgrad_pl_ok(N, H, H, Cin, Cout)) == 1

  def run(lib):
    d = dev_of(lib)
    xd, dyd = x.to(d), dy.to(d)
    ax, ay = torch.zeros(256, device=d), torch.zeros(256, device=d)
    call(lib, 'amax_partial_f32', xd, xd.numel(), ax)
    call(lib, 'amax_partial_f32', dyd, dyd.numel(), ay)
    xp = torch.zeros(int(lib.planes_bytes(N, Cin, H * H)), dtype=torch.uint8, device=d)
    yp = torch.zeros(int(lib.planes_bytes(N, Cout, H * H)), dtype=torch.uint8, device=d)
    call(lib, 'split_planes_f32', xd, N, Cin, H * H, ax, 256, xp)
    call(lib, 'split_planes_f32', dyd, N, Cout, H * H, ay, 256, yp)
    nb = max(int(lib.conv2d_wgrad_pl_ws_bytes(N, H, H, Cin, Cout)), int(lib.conv2d_wgrad_ws_bytes(Cin, 0, N, Cout, H, H, 3, 3)))
    ws = torch.full((nb // 4 + 64,), float('nan'), device=d)
    dw = dw0.clone().to(d)
    call(lib, 'conv2d_wgrad_pl_f32', xp, ax, yp, ay, dw, 0.5, ws, nb, N, H, H, Cin, Cout)
    # round 5: the caller may choose how many workgroups the K split fills (the engine: one per CU beside another stream, two alone)
    for wgs in (512, 128, 1024):
      dwv = dw0.clone().to(d)
      call(lib, 'conv2d_wgrad_pl_wgs_f32', xp, ax, yp, ay, dwv, 0.5, ws, nb, N, H, H, Cin, Cout, wgs)
      assert (dwv - dw).abs().max().item() <= 2e-6 * (dw - dw0.to(d)).abs().max().item(), wgs
    dw32 = dw0.clone().to(d)
    call(lib, 'conv2d_wgrad_f32', xd, Cin, None, 0, dyd, dw32, 0, 0.5, ws, nb, N, H, H, Cout, H, H, 3, 3, 1, 1)
    return dw.cpu(), dw32.cpu()

  (r, _), (h, h32) = run(ref_lib), run(hip_lib)
  scale = (r - dw0).abs().max().item()
  assert (h - r).abs().max().item() <= 1e-4 * scale
  assert (h - h32).abs().max().item() <= 2e-5 * scale
