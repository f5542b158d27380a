"""Practical ceiling of the 16-bit matrix pipe on this box: library GEMMs (torch.matmul -> hipBLASLt) on random fp16 / bf16
data, large square and conv-shaped problems.  The split convolutions issue three such MFMAs per fp32 product, so their
fp32-equivalent ceiling is a third of what the pipe sustains on real data (DVFS: MI355X_MICROARCH.md)."""
import torch
d = torch.device('cuda')
def timeit(f, n=20):
  for _ in range(5): f()
  torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
  s.record()
  for _ in range(n): f()
  e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e-3
for dt in (torch.float16, torch.bfloat16):
  for M, N, K in ((8192, 8192, 8192), (4096, 4096, 4096), (128, 131072, 1152), (256, 32768, 2304), (256, 131072, 2304)):
    a = torch.randn(M, K, device=d, dtype=dt); b = torch.randn(K, N, device=d, dtype=dt)
    t = timeit(lambda: torch.matmul(a, b))
    tf = 2.0 * M * N * K / t / 1e12
    print(f'{str(dt):<16} {M:6d} x {N:6d} x {K:5d}: {t * 1e6:9.1f} us  {tf:8.1f} TFLOP/s  ({tf / 2500:5.1%} of 2500; / 3 = {tf / 3:6.1f} fp32-equivalent)')
