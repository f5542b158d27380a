"""Cases for the reference's native-op surface (`op.upfirdn2d`, `op.fused_leaky_relu`, `op.FusedLeakyReLU`) and the
tensor-level up_or_down_sampling functions, run through torch.autograd -- forward, backward and double backward --
against the oracle (oracle/ref_torch.py: upfirdn2d_native restated with F.pad / F.conv2d, differentiated by torch) and
the committed reference outputs (tests/golden/ops.npz).  Shared by the CPU test (checker backend bound into the op
module) and the GPU test (HIP library)."""
import os

import numpy as np
import torch
import torch.nn.functional as F

import ref_torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ops.npz')
FIR = np.outer([1., 3., 3., 1.], [1., 3., 3., 1.]).astype(np.float32) / 64.

# (up, down, pad): the three live triples + a cropping and an odd-factor case
TRIPLES = [(1, 2, (1, 1)), (2, 1, (2, 1)), (1, 1, (2, 2)), (1, 2, (-1, 0)), (3, 2, (2, 3))]


def _close(a, b, tol=1e-5):
  a, b = a.detach().cpu().double(), b.detach().cpu().double()
  assert a.shape == b.shape, (a.shape, b.shape)
  assert (a - b).abs().max().item() <= tol * max(b.abs().max().item(), 1e-6), (a - b).abs().max().item()


def upfirdn2d_autograd(st, dev, shape=(2, 5, 12, 10)):
  """st.op.upfirdn2d forward / backward / double backward vs autograd through the oracle restatement."""
  g = torch.Generator().manual_seed(0)
  for up, down, pad in TRIPLES:
    k = torch.from_numpy(FIR * (up * up))
    x = torch.randn(shape, generator=g)
    xd = x.clone().to(dev).requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    y = st.op.upfirdn2d(xd, k.to(dev), up=up, down=down, pad=pad)
    yr = ref_torch.upfirdn2d_ref(xr, k, up=up, down=down, pad=pad)
    _close(y, yr)
    go = torch.randn(yr.shape, generator=g)
    god = go.clone().to(dev).requires_grad_(True)
    gor = go.clone().requires_grad_(True)
    gx, = torch.autograd.grad(y, xd, god, create_graph=True)
    gxr, = torch.autograd.grad(yr, xr, gor, create_graph=True)
    _close(gx, gxr)
    # double backward: d/d(go) and d/dx of <gx, v>; the operator is linear, so the x-part is zero / absent
    v = torch.randn(x.shape, generator=g)
    ggo, = torch.autograd.grad(gx, god, v.to(dev))
    ggor, = torch.autograd.grad(gxr, gor, v)
    _close(ggo, ggor)
    _close(ggo, ref_torch.upfirdn2d_ref(v, k, up=up, down=down, pad=pad))     # = the forward operator again


def upfirdn2d_golden(st, dev):
  g = np.load(GOLDEN)
  x = torch.from_numpy(g['x'])
  for name in ('down', 'up', 'pre', 'crop', 'odd'):
    up, down, p0, p1 = (int(v) for v in g[f'{name}.args'])
    xd = x.clone().to(dev).requires_grad_(True)
    y = st.op.upfirdn2d(xd, torch.from_numpy(g[f'{name}.k']).to(dev), up=up, down=down, pad=(p0, p1))
    _close(y, torch.from_numpy(g[f'{name}.y']))
    y.backward(torch.from_numpy(g[f'{name}.go']).to(dev))
    _close(xd.grad, torch.from_numpy(g[f'{name}.gx']))


def resampling_wrappers_golden(st, dev):
  """models.up_or_down_sampling tensor-level functions against the reference's own outputs, and their gradients
  against autograd through the oracle restatement."""
  uds = st.models.up_or_down_sampling
  g = np.load(GOLDEN)
  x = torch.from_numpy(g['x'])
  w = torch.from_numpy(g['uds.w'])
  got = {
    'uds.up': uds.upsample_2d(x.to(dev), (1, 3, 3, 1), factor=2),
    'uds.down': uds.downsample_2d(x.to(dev), (1, 3, 3, 1), factor=2),
    'uds.conv_down': uds.conv_downsample_2d(x.to(dev), w.to(dev), k=(1, 3, 3, 1)),
    'uds.naive_up': uds.naive_upsample_2d(x.to(dev)),
    'uds.naive_down': uds.naive_downsample_2d(x.to(dev)),
  }
  for k, v in got.items():
    _close(v, torch.from_numpy(g[k]), 2e-5)
  # gradients of the fused FIR + strided conv wrt x and w
  gen = torch.Generator().manual_seed(3)
  xb = torch.randn(3, 6, 16, 16, generator=gen)
  wb = torch.randn(7, 6, 3, 3, generator=gen) * 0.2
  xd, wd = xb.clone().to(dev).requires_grad_(True), wb.clone().to(dev).requires_grad_(True)
  xr, wr = xb.clone().requires_grad_(True), wb.clone().requires_grad_(True)
  y = uds.conv_downsample_2d(xd, wd, k=(1, 3, 3, 1))
  yr = ref_torch.conv_downsample_2d(xr, wr, (1, 3, 3, 1))
  _close(y, yr, 2e-5)
  go = torch.randn(yr.shape, generator=gen)
  y.backward(go.to(dev))
  yr.backward(go)
  _close(xd.grad, xr.grad, 2e-5)
  _close(wd.grad, wr.grad, 2e-5)


def _flr_ref(x, b, slope, scale):
  return F.leaky_relu(x + b.view(1, -1, *([1] * (x.dim() - 2))), slope) * scale


def fused_leaky_relu_autograd(st, dev):
  g = np.load(GOLDEN)
  x, b = torch.from_numpy(g['x']), torch.from_numpy(g['flr.bias'])
  _close(st.op.fused_leaky_relu(x.to(dev), b.to(dev), 0.2, 2 ** 0.5), torch.from_numpy(g['flr.y']))
  gen = torch.Generator().manual_seed(5)
  for shape, slope, scale in (((3, 4, 6, 5), 0.2, 2 ** 0.5), ((5, 7), 0.1, 1.0), ((2, 3, 4, 2, 3), 0.3, 0.5)):
    xs = torch.randn(shape, generator=gen)
    bs = torch.randn(shape[1], generator=gen)
    xd, bd = xs.clone().to(dev).requires_grad_(True), bs.clone().to(dev).requires_grad_(True)
    xr, br = xs.clone().requires_grad_(True), bs.clone().requires_grad_(True)
    y = st.op.fused_leaky_relu(xd, bd, slope, scale)
    yr = _flr_ref(xr, br, slope, scale)
    _close(y, yr)
    go = torch.randn(shape, generator=gen)
    god, gor = go.clone().to(dev).requires_grad_(True), go.clone().requires_grad_(True)
    gx, gb = torch.autograd.grad(y, (xd, bd), god, create_graph=True)
    gxr, gbr = torch.autograd.grad(yr, (xr, br), gor, create_graph=True)
    _close(gx, gxr)
    _close(gb, gbr)
    # grad-grad: differentiate <gx, v> + <gb, u> wrt the incoming gradient
    v, u = torch.randn(shape, generator=gen), torch.randn(shape[1], generator=gen)
    ggo, = torch.autograd.grad((gx * v.to(dev)).sum() + (gb * u.to(dev)).sum(), god)
    ggor, = torch.autograd.grad((gxr * v).sum() + (gbr * u).sum(), gor)
    _close(ggo, ggor)
  # the module: state_dict surface and a parameter gradient
  m = st.op.FusedLeakyReLU(4).to(dev)
  assert list(m.state_dict()) == ['bias']
  with torch.no_grad():
    m.bias.copy_(torch.tensor([0.5, -0.25, 0.0, 1.0]))
  xs = torch.randn(2, 4, 3, 3, generator=gen)
  out = m(xs.to(dev))
  _close(out, _flr_ref(xs, m.bias.detach().cpu(), 0.2, 2 ** 0.5))
  out.sum().backward()
  mask = (_flr_ref(xs, m.bias.detach().cpu(), 0.2, 1.0) > 0).float()
  _close(m.bias.grad, ((mask + (1 - mask) * 0.2) * 2 ** 0.5).sum((0, 2, 3)))


def other_dtypes(st, dev):
  """The half and double entry points of the two native ops (the reference dispatches AT_DISPATCH_FLOATING_TYPES_AND_HALF,
  op/upfirdn2d_kernel.cu:311, op/fused_bias_act_kernel.cu:77): forward and input gradient against the reference's own
  outputs (tests/golden/ops.npz, fp32) and the oracle restatement in double."""
  g = np.load(GOLDEN)
  x = torch.from_numpy(g['x'])
  for dtype, tol in ((torch.float64, 1e-6), (torch.float16, 4e-3)):
    for name in ('down', 'up', 'pre', 'crop', 'odd'):
      up, down, p0, p1 = (int(v) for v in g[f'{name}.args'])
      xd = x.clone().to(dev, dtype).requires_grad_(True)
      y = st.op.upfirdn2d(xd, torch.from_numpy(g[f'{name}.k']).to(dev), up=up, down=down, pad=(p0, p1))
      assert y.dtype == dtype
      _close(y, torch.from_numpy(g[f'{name}.y']), tol)
      y.backward(torch.from_numpy(g[f'{name}.go']).to(dev, dtype))
      assert xd.grad.dtype == dtype
      _close(xd.grad, torch.from_numpy(g[f'{name}.gx']), tol)
    b = torch.from_numpy(g['flr.bias'])
    xd = x.clone().to(dev, dtype).requires_grad_(True)
    bd = b.clone().to(dev, dtype).requires_grad_(True)
    y = st.op.fused_leaky_relu(xd, bd, 0.2, 2 ** 0.5)
    assert y.dtype == dtype
    _close(y, torch.from_numpy(g['flr.y']), tol)
    xr, br = x.clone().double().requires_grad_(True), b.clone().double().requires_grad_(True)
    go = torch.randn(x.shape, generator=torch.Generator().manual_seed(2))
    y.backward(go.to(dev, dtype))
    _flr_ref(xr, br, 0.2, 2 ** 0.5).backward(go.double())
    _close(xd.grad, xr.grad, tol)
    _close(bd.grad, br.grad, 4 * tol)
  # double precision really is double: a value fp32 cannot hold survives the pass-through FIR
  one = torch.tensor([[1.0]], dtype=torch.float64)
  v = torch.full((1, 1, 4, 4), 1.0 + 2.0 ** -40, dtype=torch.float64)
  out = st.op.upfirdn2d(v.to(dev), one.to(dev))
  assert torch.equal(out.cpu(), v)
