"""HIP-event timing of individual kernel launches, on the stream they are launched on.

``KernelTimer`` is attached to an executor (``executor.profiler = KernelTimer()``); the graph ops then
bracket every contraction launch with a pair of events recorded on torch's current stream -- the same
stream the C-ABI launch uses -- so ``elapsed_time`` is that kernel's duration.  bench.py uses it to report
the roofline of the dominant kernel from measurements taken inside the timed region.
"""
import collections

import torch


class KernelTimer:
  def __init__(self):
    self.records = []          # (kind, flops, start_event, end_event)

  def launch(self, kind, flops, fn, args):
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    s.record()
    fn(*args)
    e.record()
    self.records.append((kind, flops, s, e))

  def summary(self):
    """kind -> dict(count, total_ms, avg_us, flops_per_launch, tflops).  Call after a device sync."""
    agg = collections.OrderedDict()
    for kind, flops, s, e in self.records:
      a = agg.setdefault(kind, dict(count=0, total_ms=0.0, flops=0.0))
      a['count'] += 1
      a['total_ms'] += s.elapsed_time(e)
      a['flops'] += flops
    for a in agg.values():
      a['avg_us'] = 1e3 * a['total_ms'] / a['count']
      a['flops_per_launch'] = a['flops'] / a['count']
      a['tflops'] = a['flops'] / (a['total_ms'] * 1e-3) / 1e12 if a['total_ms'] > 0 else 0.0
    return agg

  def reset(self):
    self.records = []
