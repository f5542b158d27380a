"""Accuracy of the convolution paths against float64 on the device (run on a GPU box):
split = the scratch-given path of stk_conv2d_fwd_f32 (fp16 two-way split, conv_x2.h), f32 = f32-input MFMA (ws = NULL),
wgrad = stk_conv2d_wgrad_f32 (fp16 two-way split of both operands) with dy = a gradient-like tensor."""
import os, sys, importlib, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = importlib.import_module('soft-truncation_amd.engine.lib').load()
dev = torch.device('cuda:0')
stream = torch.cuda.current_stream(dev).cuda_stream
g = torch.Generator().manual_seed(0)
rows = []
for N, Cin, Cout, H, xs, wsc, spread, what in [
    (24, 128, 128, 32, 1.0, 1 / 34., 0.0, 'unit-scale activations'),
    (48, 256, 256, 16, 1.0, 1 / 48., 0.0, '256 channels'),
    (24, 128, 128, 32, 1e-5, 1 / 34., 4.0, 'gradient-like: 1e-5, images spread over 4 decades'),
    (24, 128, 128, 32, 50.0, 1e-3, 0.0, 'large activations, small weights'),
    (24, 512, 256, 16, 1.0, 1 / 68., 0.0, 'K = 4608')]:
  x = torch.randn(N, Cin, H, H, generator=g) * xs
  if spread:
    x = x * (10.0 ** (-spread * torch.arange(N).float() / (N - 1))).view(N, 1, 1, 1)
  x = x.to(dev)
  w = (torch.randn(Cout, Cin, 3, 3, generator=g) * wsc).to(dev)
  ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
  shape = (Cin, 0, N, H, H, Cout, 3, 3, 1, 1)
  nb = int(lib.conv2d_fwd_ws_bytes(*shape))
  ws = torch.empty(nb // 4 + 64, device=dev)
  errs = {}
  for name in ('split', 'f32'):
    y = torch.empty(N, Cout, H, H, device=dev)
    lib.conv2d_fwd_f32(x.data_ptr(), Cin, None, 0, w.data_ptr(), 0, None, None, 0, None, 1.0, y.data_ptr(), N, H, H, Cout, H, H,
                       3, 3, 1, 1, ws.data_ptr() if name == 'split' else None, nb if name == 'split' else 0, stream)
    torch.cuda.synchronize()
    d = (y.double() - ref).abs()
    # max-norm error, and the worst per-image error relative to that image's own maximum
    per_img = (d.flatten(1).max(1).values / ref.abs().flatten(1).max(1).values).max()
    errs[name] = (float(d.max() / ref.abs().max()), float(per_img))
  # weight gradient: dy with per-image magnitudes spread over three decades, as a loss-weighted batch has
  dy = (torch.randn(N, Cout, H, H, generator=g) * 1e-4 * (10.0 ** (-3.0 * torch.arange(N).float() / (N - 1))).view(N, 1, 1, 1)).to(dev)
  xd = x.double().requires_grad_(False)
  wd = w.double().requires_grad_(True)
  (torch.nn.functional.conv2d(xd, wd, padding=1) * dy.double()).sum().backward()
  nbw = int(lib.conv2d_wgrad_ws_bytes(Cin, 0, N, Cout, H, H, 3, 3))
  wws = torch.empty(nbw // 4 + 64, device=dev)
  dw = torch.zeros_like(w)
  lib.conv2d_wgrad_f32(x.data_ptr(), Cin, None, 0, dy.data_ptr(), dw.data_ptr(), 0, 1.0, wws.data_ptr(), wws.numel() * 4,
                       N, H, H, Cout, H, H, 3, 3, 1, 1, stream)
  torch.cuda.synchronize()
  werr = float((dw.double() - wd.grad).abs().max() / wd.grad.abs().max())
  rows.append(f'{what:<52} ' + '  '.join(f'{k}: {v[0]:.2e} (per image {v[1]:.2e})' for k, v in errs.items()) + f'  wgrad: {werr:.2e}')
  print(rows[-1], flush=True)
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
open(os.path.join(ROOT, 'gpurun_out', 'split_accuracy.txt'), 'w').write('\n'.join(rows) + '\n')
