#!/usr/bin/env python
"""Development aid: cycle stamps of workgroup 0 / wave 0 of the fused attention forward (stk_attention_fwd_debug)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import soft_truncation_amd as st
import os
lib = st.engine.lib.load_path(os.environ['STK_LIB']) if os.environ.get('STK_LIB') else st.engine.lib.load()
B, C, T = 128, 256, 256
d = torch.device('cuda:0')
q, k, v = (torch.randn(B, C, T, device=d) for _ in range(3))
o, lse, rec = torch.empty(B, C, T, device=d), torch.empty(B, T, device=d), torch.empty(1024, device=d)
lib.attention_fwd_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), C * T, o.data_ptr(), lse.data_ptr(), rec.data_ptr(), B, C, T, C ** -0.5, 0)
dbg = torch.zeros(16, dtype=torch.int64, device=d)
fn = lib._cdll.stk_attention_fwd_debug
fn.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 3 + [ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]
for _ in range(3):
  fn(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), rec.data_ptr(), B, C, T, C ** -0.5, dbg.data_ptr(), 0)
torch.cuda.synchronize()
t = dbg.cpu().tolist()
names = ['start', 'scales', 'phase A', 'softmax', 'put_bm + B prologue', 'B loop', 'epilogue']
for i in range(1, 7):
  print(f'{names[i]:<24} {t[i] - t[i - 1]:8d} cycles')
print('total', t[6] - t[0])
