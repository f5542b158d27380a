#!/usr/bin/env python
"""Census of the planned graph of a workload (no GPU needed): every op class with its shapes, launch labels and
algorithmic FLOPs / bytes, grouped.  usage: python tools/census.py [workload] [batch]"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import soft_truncation_amd as st
from importlib import import_module

graph_mod = import_module('soft-truncation_amd.engine.graph')


def main():
  wl = sys.argv[1] if len(sys.argv) > 1 else 'cifar10_ddpmpp_nll_st'
  B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
  cfg = getattr(st.configs, wl)()
  cfg.device = torch.device('cpu')
  sde = st.sde_lib.get_sde(cfg, None)
  lib = st.engine.lib.load()
  net = st.models.ncsnpp.NCSNpp(cfg, sde)
  H = cfg.data.image_size
  flat = import_module('soft-truncation_amd.engine.flat').FlatParams(list(net.parameters()), torch.device('cpu'),
                                                                     groups=net._flat_groups())
  g = graph_mod.Graph(flat, lib)
  out = net._emit(g, B, H, H, False)
  g.finalize(out, lib)
  rows = collections.OrderedDict()
  for op in g.ops:
    if isinstance(op, graph_mod.Conv):
      key = ('Conv', op.C1, op.C2, op.Cout, op.H, op.OH, op.KH, op.stride, 'pl' if op.pl_fwd else '-',
             'pd' if op.pl_dgrad else '-', 'pw' if op.pl_wgrad else '-', op._kind(lib, 'fwd'), op._kind(lib, 'dgrad'),
             op._kind(lib, 'wgrad'), 'res' if op.res is not None else '-', 'temb' if op.temb is not None else '-')
      r = rows.setdefault(key, [0, 0.0])
      r[0] += 1
      r[1] += op.flops
    else:
      shp = getattr(op, 'y', None)
      key = (type(op).__name__,) + (tuple(shp.shape) if shp is not None else ())
      if isinstance(op, graph_mod.GroupNormAct):
        key = key + (op.C1, op.C2, 'fused' if op.fused else '-', 'plmaker' if op.y.pl_maker is op else '-',
                     'f32fwd' if op.y.f32_fwd else '-', 'f32bwd' if op.y.f32_bwd else '-')
      r = rows.setdefault(key, [0, 0.0])
      r[0] += 1
  for k, (n, fl) in rows.items():
    print(f'{n:4d} {fl / 1e9:9.2f} GF  ', ' '.join(str(x) for x in k))
  print('ops', len(g.ops), 'act MB', g.act_size * 4 / 1e6, 'planes MB', g.pl_bytes / 1e6)


if __name__ == '__main__':
  main()
