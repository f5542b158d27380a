"""Third-party victims of the gfx950 packed-fp32 hazard: torch's OWN kernels that contain the affected instruction form
(profiles/r04_torch_hip_pk_scan.json: 782 kernels of libtorch_hip.so, 9261 instructions) run on the current stream while another
stream runs the library's AccVGPR-MFMA weight gradient (the neighbour of tests/_model_cases.py).  Every result is compared
bit for bit with the quiet result of the same call.   python tools/_probe/torch_victim.py [--rounds 150]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
  sys.path.insert(0, p)
import torch
from importlib import import_module
import soft_truncation_amd as st
from _model_cases import _MfmaNeighbour
ex = import_module('soft-truncation_amd.engine.executor')

ap = argparse.ArgumentParser()
ap.add_argument('--rounds', type=int, default=150)
args = ap.parse_args()
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
lib = st.engine.lib.load()
main = torch.cuda.current_stream(dev)
nb = _MfmaNeighbour(lib, dev)
for _ in range(16):                       # a stream that really runs beside the current one (see checked_side_stream)
  s = torch.cuda.Stream(dev)
  if ex._overlap_ratio(main, s) < 1.5:
    nb.stream = s
    break
print('neighbour stream overlap ratio', round(ex._overlap_ratio(main, nb.stream), 2), flush=True)
g = torch.Generator().manual_seed(3)
x = torch.randn(1 << 22, generator=g).to(dev)
y = (torch.rand(1 << 22, generator=g) * 2 + 0.5).to(dev)
xs = x.view(4096, 1024)
cases = {
  'vector_norm (reduce_kernel<..NormOps<float>>)': lambda: torch.linalg.vector_norm(x),
  'norm rows (reduce_kernel, dim=1)': lambda: torch.linalg.vector_norm(xs, dim=1),
  'pow(tensor, tensor)': lambda: torch.pow(y, x),
  'mish backward': lambda: torch.ops.aten.mish_backward(x, y),
  'cumsum (rocprim scan)': lambda: torch.cumsum(xs, dim=1),
  'add (control: plain elementwise)': lambda: torch.add(x, y),
}
for name, fn in cases.items():
  try:
    ref = fn().clone()
    torch.cuda.synchronize()
    quiet_bad = sum(int(not torch.equal(fn(), ref)) for _ in range(50))
    bad = calls = 0
    worst = 0.0
    for r in range(args.rounds):
      nb.launch(30)
      for _ in range(12):
        out = fn()
        calls += 1
        if not torch.equal(out, ref):
          bad += 1
          d = (out.double() - ref.double()).abs().max().item() / max(ref.double().abs().max().item(), 1e-30)
          worst = max(worst, d)
      torch.cuda.synchronize()
    print(f'{name:48s} quiet: {quiet_bad} of 50 differ; beside the MFMA neighbour: {bad} of {calls} differ (worst max-rel {worst:.3e})', flush=True)
  except Exception as e:
    print(f'{name:48s} error {e!r}'[:200], flush=True)

# control: the library's own GroupNorm backward compiled WITH the SLP vectoriser (tools/_probe/build/libstk_gnslp.so, if present)
# under exactly this neighbour -- the case that diverged in round 3
slp = os.path.join(ROOT, 'tools', '_probe', 'build', 'libstk_gnslp.so')
if os.path.exists(slp):
  LIBMOD = import_module('soft-truncation_amd.engine.lib')
  slib = LIBMOD.load_path(slp)
  sys.path.insert(0, os.path.join(ROOT, 'tests'))
  from _util import call
  N, C, H, G = 96, 192, 16, 32
  gx = torch.randn(N, C, H, H, generator=g).to(dev); gdy = torch.randn(N, C, H, H, generator=g).to(dev)
  gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
  mean, rstd = torch.zeros(N * G, device=dev), torch.ones(N * G, device=dev)
  dgm, dbt = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
  ws = torch.empty(int(slib.gn_ws_bytes(N, C, H * H, G)) // 4 + 64, device=dev)
  def gn(l):
    dx = torch.empty_like(gx)
    call(l, 'gn_bwd_f32', gdy, gx, C, None, 0, gam, bet, mean, rstd, dx, 0.0, None, 0.0, dgm, dbt, ws, N, H * H, G, 1, 0.0, 1, None)
    return dx
  for tag, l in (('SLP build', slib), ('shipped library', lib)):
    ref = gn(l).clone(); torch.cuda.synchronize()
    bad = calls = 0
    for r in range(args.rounds):
      nb.launch(30)
      for _ in range(12):
        calls += 1
        bad += int(not torch.equal(gn(l), ref))
      torch.cuda.synchronize()
    print(f'GroupNorm backward, {tag:16s} beside the MFMA neighbour: {bad} of {calls} differ', flush=True)
