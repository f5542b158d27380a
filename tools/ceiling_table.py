#!/usr/bin/env python
"""DESIGN.md section 8, "Ceiling and gap", from a bench.py detail file: per kernel family the algorithmic FLOPs per step, the time the
family takes now (bench.py's event brackets: one-stream eager steps, kernel + its slab reduce), the time it would take at the best rate
any contraction kernel of this design reaches INSIDE the step (roofline_best), and the difference.

    python tools/ceiling_table.py profiles/r06_bench_detail.json
"""
import json
import sys

FAMILIES = [
  ('3x3 forward, large maps (halo GEMM / LDS-DMA GEMM)', ('conv3x3.fwd.x2p.h', 'conv3x3.fwd.x2p$')),
  ('3x3 data gradient, large maps (same kernels)', ('conv3x3.dgrad.x2p.h', 'conv3x3.dgrad.x2p$')),
  ('3x3 forward + data gradient, small maps (K split + slab sum)', ('conv3x3.fwd.x2p.k', 'conv3x3.dgrad.x2p.k')),
  ('3x3 weight gradient, 32- / 16-wide maps', ('conv3x3.wgrad.x2p.w32', 'conv3x3.wgrad.x2p.w16')),
  ('3x3 weight gradient, 8- / 4-wide maps', ('conv3x3.wgrad.x2p.w8', 'conv3x3.wgrad.x2p.w4')),
  ('1x1 layers (shortcuts, NIN, q / k / v), all directions', ('conv1x1.',)),
  ('attention core, forward + backward', ('attention.',)),
  ('thin-side layers (stem, head, 3-channel pyramids)', ('.thin',)),
]


def match(name, pats):
  for p in pats:
    if p.endswith('$'):
      if name == p[:-1]:
        return True
    elif p in name:
      return True
  return False


def main(path, workload=None):
  d = json.load(open(path))
  if workload:
    d = d['workloads'][workload]
  k = d['kernels']
  best = d['roofline_best']
  rate = best['achieved']
  print(f'best in-step contraction rate: {best["kernel"]} {rate:.1f} TFLOP/s = {best["frac"]:.3f} of {best["peak"]:.0f}; '
        f'step {d["ms_per_step"]:.2f} ms, {d["value"]:.1f} images/s')
  print('| kernel family | TFLOP per step | ms now (one stream) | TFLOP/s now | ms at the best in-step rate | gap ms |')
  print('|---|---|---|---|---|---|')
  seen = set()
  tt = tf = tfl = 0.0
  for title, pats in FAMILIES:
    names = [n for n in k if n not in seen and match(n, pats)]
    seen.update(names)
    if not names:
      continue
    t = sum(k[n]['total_ms_per_step'] for n in names)
    f = sum(k[n]['tflops'] * k[n]['total_ms_per_step'] * 1e-3 for n in names)
    fl = f / rate * 1e3
    tt, tf, tfl = tt + t, tf + f, tfl + fl
    print(f'| {title} | {f:.3f} | {t:.2f} | {f / t * 1e3:.0f} | {fl:.2f} | {t - fl:+.2f} |')
  rest = [n for n in k if n not in seen]
  if rest:
    t = sum(k[n]['total_ms_per_step'] for n in rest)
    f = sum(k[n]['tflops'] * k[n]['total_ms_per_step'] * 1e-3 for n in rest)
    tt, tf, tfl = tt + t, tf + f, tfl + f / rate * 1e3
    print(f'| other ({", ".join(rest)}) | {f:.3f} | {t:.2f} | {f / max(t, 1e-9) * 1e3:.0f} | {f / rate * 1e3:.2f} | {t - f / rate * 1e3:+.2f} |')
  print(f'| **all contractions** | **{tf:.2f}** | **{tt:.2f}** | {tf / tt * 1e3:.0f} | **{tfl:.2f}** | {tt - tfl:+.2f} |')


if __name__ == '__main__':
  main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
