"""Times the pieces of one sampler network evaluation (diagnostic)."""
import sys, time, torch, numpy as np
sys.path.insert(0, '.')
import soft_truncation_amd as st
cfg = st.configs.get_config('cifar10_ddpmpp_nll_st'); cfg.device = torch.device('cuda')
sde = st.sde_lib.get_sde(cfg, None)
model = st.models.utils.create_model(cfg, sde)
score_fn = st.models.utils.get_score_fn(cfg, sde, model, train=False, continuous=True)
x = torch.randn(128, 3, 32, 32, device='cuda'); t = torch.full((128,), 0.5, device='cuda')
def timeit(f, n=10):
  for _ in range(3): f()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(n): f()
  torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
  print('score_fn fwd  ms', timeit(lambda: score_fn(x, t)))
  rsde = sde.reverse(score_fn, probability_flow=True, lambda_=0.)
  print('rsde.sde      ms', timeit(lambda: rsde.sde(x, t)[0]))
  def roundtrip():
    xn = x.detach().cpu().numpy().reshape(-1).astype(np.float64)
    return torch.from_numpy(xn.reshape(x.shape)).to('cuda').type(torch.float32)
  print('host roundtrip ms', timeit(roundtrip))
