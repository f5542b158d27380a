"""Data parallelism: one process per GPU, one bucketed gradient all-reduce per step over RCCL/xGMI.

The reference's only multi-GPU mechanism is ``torch.nn.DataParallel`` (models/utils.py:94):
every forward re-broadcasts all 247 MB of parameters from GPU 0 and every backward reduces the
gradients back to GPU 0 from Python threads.  MI355X-first design instead:

* each rank owns a full replica and a disjoint slice of the global batch; GroupNorm is
  per-sample, so the only exchange in a step is the gradient average;
* gradients already live in ONE flat fp32 buffer (engine/flat.py), so the exchange is a few large
  contiguous all-reduces (``bucket_mb`` each, default 64 MB: with 7 point-to-point xGMI links per
  GPU the ring is per-link bound, so few large messages beat many small ones), issued
  asynchronously and waited once before the clip/Adam kernels;
* ``t_min`` is one host scalar per step drawn from numpy's global stream (losses.py:284), so all
  ranks seed numpy identically (``seed_everything``) and use the same truncation bound, as the
  replicas of DataParallel did; torch's generators (t, z, dropout) are seeded per rank.

Works with any backend ``torch.distributed`` offers: "nccl" (= RCCL on ROCm) on GPUs, "gloo"
for the CPU tests.
"""
import numpy as np
import torch
import torch.distributed as dist


# STK_DDP_FORCE=1: run the exchange (segmented backward, bucket all-reduces, averaging) in a process group of ONE rank
# too -- the fixed per-step cost every rank of a multi-GPU run pays, measurable on a single GPU (bench.py --force-exchange)
FORCE_SINGLE_RANK = False
# bench.py's exchange proxy (one rank): a one-rank all-reduce moves nothing, so nothing would stand beside the backward.  With
# PROXY_TRAFFIC every bucket handed over also travels once through the communicator's stream as an out-of-place collective
# (a one-rank all-gather = a device copy of the bucket: read + write of its bytes, about what a ring step's reduce-copy kernels
# move per bucket), so the backward really shares HBM and the chip with the exchange stream.  Never set outside the proxy.
PROXY_TRAFFIC = False
_proxy_scratch = {}


def is_distributed():
  import os
  if not (dist.is_available() and dist.is_initialized()):
    return False
  return dist.get_world_size() > 1 or FORCE_SINGLE_RANK or os.environ.get('STK_DDP_FORCE', '0') == '1'


def world_size():
  return dist.get_world_size() if is_distributed() else 1


def rank():
  return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def seed_everything(seed):
  """numpy: identical on all ranks (shared t_min stream); torch: offset by rank."""
  np.random.seed(seed)
  torch.manual_seed(seed + rank())
  if torch.cuda.is_available():
    torch.cuda.manual_seed(seed + rank())


def bucket_ranges(n, bucket_elems):
  """Split [0, n) into contiguous buckets of at most ``bucket_elems`` elements."""
  bucket_elems = max(int(bucket_elems), 1)
  return [(s, min(s + bucket_elems, n)) for s in range(0, n, bucket_elems)]


def allreduce_flat_(flat_grad, n, bucket_mb=64.0, average=True, group=None):
  """In-place sum (or mean) of ``flat_grad[:n]`` across ranks, bucketed and asynchronous."""
  if not is_distributed():
    return 0
  ws = dist.get_world_size(group)
  handles = []
  ranges = bucket_ranges(n, int(bucket_mb * (1 << 20) / 4))
  for s, e in ranges:
    handles.append(dist.all_reduce(flat_grad[s:e], op=dist.ReduceOp.SUM, group=group, async_op=True))
  for h in handles:
    h.wait()
  if average:
    flat_grad[:n].div_(ws)
  return len(ranges)


class OverlappedExchange:
  """Gradient all-reduce overlapped with the backward pass of the LAST micro-batch of a step.

  The engine's backward is a fixed launch sequence over one flat gradient buffer (engine/executor.py).  Armed on an
  executor, it cuts that sequence where buckets of the buffer become final -- from the top of the buffer down, which
  is the order the backward finishes them -- and calls :meth:`ready` between the cuts; `ready` starts the bucket's
  asynchronous all-reduce, which RCCL runs on its own stream behind the launches made so far while the next segment
  computes.  :meth:`finish` (from ``sync_gradients``) reduces whatever was not covered, waits once, and averages.
  Replaces the reference's ``torch.nn.DataParallel`` gather (models/utils.py:94) -- with the same sums: a bucket is
  only handed over after its last writer, so the result equals the exchange-after-backward path bit for bit."""

  def __init__(self, executor, bucket_mb=64.0, group=None):
    self.ex, self.group = executor, group
    self.bucket_elems = int(bucket_mb * (1 << 20) / 4)
    self.handles, self.covered, self.flat = [], [], None

  def arm(self):
    self.flat = self.ex.ensure_flat()
    self.handles, self.covered = [], []
    self.ex.grad_bucket_elems = self.bucket_elems
    self.ex.grad_hook = self.ready
    return self

  def ready(self, lo, hi):
    if hi > lo:
      self.handles.append(dist.all_reduce(self.flat.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
      self.covered.append((lo, hi))
      if PROXY_TRAFFIC and dist.get_world_size(self.group) == 1:
        g = self.flat.grad
        buf = _proxy_scratch.get(g.device)
        if buf is None or buf.numel() < self.flat.n_train:
          buf = _proxy_scratch[g.device] = torch.empty(self.flat.n_train, dtype=g.dtype, device=g.device)
        self.handles.append(dist.all_gather_into_tensor(buf[lo:hi], g[lo:hi], group=self.group, async_op=True))

  def finish(self, average=True):
    """Reduce the ranges no segment handed over (a backward that never ran, unused parameters), wait, average."""
    self.ex.grad_hook = None
    flat, n = self.flat, self.flat.n_train
    pos = 0
    for lo, hi in sorted(self.covered):
      if lo > pos:
        self.ready(pos, lo)
      pos = max(pos, hi)
    if pos < n:
      self.ready(pos, n)
    for h in self.handles:
      h.wait()
    if average:
      flat.grad[:n].div_(dist.get_world_size(self.group))
    nb = len(self.handles)
    self.handles, self.covered = [], []
    return nb


_armed = {}      # id(FlatParams) -> OverlappedExchange armed for the current step
# Set when a step raised on this rank WHILE bucket all-reduces were in flight (disarm_overlap(wait=False)): the dropped
# collectives may still be writing into flat.grad, and the ranks no longer agree on which buckets were issued, so the next
# collective could pair with the wrong peer call or hang with no pointer to the cause.  Under data parallelism such an
# exception is fatal for the process group: every later exchange raises this message instead.
_poisoned = None


def _check_poison():
  if _poisoned is not None:
    raise RuntimeError('gradient exchange unusable: ' + _poisoned + ' -- destroy and re-create the process group (all ranks) '
                       'before training on')


def _default_bucket_mb():
  import os
  return float(os.environ.get('STK_DDP_BUCKET_MB', '64'))


def arm_overlap(model, bucket_mb=None):
  """Called by the training step right before the backward of its last micro-batch (losses.get_step_fn): from then on
  finished gradient buckets are all-reduced while the backward is still running.  No-op in a single process and for
  models that are not on the engine."""
  if not is_distributed():
    return None
  _check_poison()
  net = getattr(model, 'module', model)
  engine = getattr(net, 'engine', None)
  if engine is None:
    return None
  ex = engine()
  x = OverlappedExchange(ex, _default_bucket_mb() if bucket_mb is None else bucket_mb).arm()
  _armed[id(x.flat)] = x
  return x


def begin_step(model):
  """Start of a training step: evaluations of EARLIER steps that never ran a backward (a grad-mode output the caller kept
  and did not differentiate) no longer count as pending -- otherwise a single forgotten output would silently turn every
  later step's overlapped exchange into a plain post-backward all-reduce (engine/executor.py `_awaiting`)."""
  net = getattr(model, 'module', model)
  engine = getattr(net, 'engine', None)
  if engine is not None:
    engine()._awaiting.clear()


def disarm_overlap(model, wait=True):
  """Drop whatever :func:`arm_overlap` left armed for `model`: the executor's hook is cleared and outstanding handles are
  waited for, so the next step starts clean.  `wait=False` (a step that RAISED on this rank): the handles are dropped
  without waiting -- the peers may never issue the matching collective, and a wait here would hang the process and hide
  the exception that is about to surface.  If any were in flight the exchange is marked unusable (`_poisoned`): a caller
  that catches the exception and keeps training gets a clear error from the next arm_overlap / sync_gradients instead of
  mismatched collectives later (a new process group -- dist.destroy_process_group + init on all ranks -- clears it through
  `reset_poison`)."""
  net = getattr(model, 'module', model)
  engine = getattr(net, 'engine', None)
  if engine is None:
    return
  ex = engine()
  ex.grad_hook = None
  x = _armed.pop(id(ex.flat), None) if ex.flat is not None else None
  if x is not None:
    if wait:
      for h in x.handles:
        h.wait()
    elif x.handles:
      global _poisoned
      _poisoned = (f'a training step raised on rank {rank()} with {len(x.handles)} bucket all-reduce(s) in flight, which '
                   f'were dropped without waiting (their peers may never issue the matching calls)')
    x.handles, x.covered = [], []


def reset_poison():
  """After the process group has been re-created on every rank."""
  global _poisoned
  _poisoned = None


# ---- does the exchange really run BESIDE the backward? ---------------------------------------------------------------------
# HIP serves its streams from a few hardware queues; two streams on one queue run strictly one after the other (engine/
# executor.py: checked_side_stream, profiles/r04_side_stream_queue.txt).  ProcessGroupNCCL draws the stream its collectives run
# on from the same torch pool as the engine's side stream, and it offers no handle on it -- a communicator whose stream landed
# on the launch stream's queue (or the side stream's) would serialise every "overlapped" bucket all-reduce behind the
# backward it is meant to hide under, invisibly on one GPU and at the cost of the whole design on eight.  The probe below
# needs no handle: it keeps one stream busy with a spin kernel and asks whether a collective issued meanwhile (from a third,
# idle stream) completes before the spin ends.

def _spin_cycles(device, ms):
  """torch.cuda._sleep argument for about `ms` milliseconds on this device (calibrated once per call)."""
  s = torch.cuda.current_stream(device)
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  torch.cuda._sleep(1000)
  a.record(s)
  torch.cuda._sleep(2_000_000)
  b.record(s)
  b.synchronize()
  per_ms = 2_000_000 / max(a.elapsed_time(b), 1e-3)
  return int(per_ms * ms)


def collective_runs_beside(busy, device, group=None, numel=8 << 20, spin_ms=40.0, trials=2):
  """True when a collective of the process group completes WHILE `busy` (a torch stream) is occupied by a spin kernel, i.e.
  the communicator's stream is served by another hardware queue.  Every rank of the group must call this together (the probe
  is an all-gather of `numel` floats -- with one rank RCCL turns it into a device copy on the communicator's stream, with
  several into its usual kernels).  Returns (beside, seconds the collective took to complete, spin seconds).

  With several ranks the collective completes only after the SLOWEST peer has issued it, so rank skew longer than the spin
  reads as "not beside" (the other error, a false "beside", cannot happen: a collective on the busy stream's queue never
  finishes before the spin).  Hence: the ranks meet at a barrier right before the timed section, the spin grows with the world
  size, and a "not beside" verdict is re-tried (`trials`; the verdict is the best one -- every rank runs every trial, so the
  collectives stay matched)."""
  ws = dist.get_world_size(group)
  best = None
  for _ in range(max(1, trials) if ws > 1 else 1):
    r = _collective_runs_beside_once(busy, device, group, numel, spin_ms * (1.0 + 0.25 * (ws - 1)))
    if best is None or (r[0] and not best[0]):
      best = r
  return best


def _collective_runs_beside_once(busy, device, group, numel, spin_ms):
  import time
  from .executor import _overlap_ratio
  ws = dist.get_world_size(group)
  src = torch.ones(numel, device=device)
  dst = torch.empty(numel * ws, device=device)
  issue = None
  for _ in range(16):                       # an idle stream that itself runs beside `busy`
    cand = torch.cuda.Stream(device)
    if cand.cuda_stream != busy.cuda_stream and _overlap_ratio(busy, cand) < 1.5:
      issue = cand
      break
  if issue is None:
    raise RuntimeError('no pooled stream runs beside the given one: cannot probe the exchange stream')
  with torch.cuda.stream(issue):            # first use creates the communicator's stream / connections: not timed
    dist.all_gather_into_tensor(dst, src, group=group)
  torch.cuda.synchronize(device)
  cycles = _spin_cycles(device, spin_ms)
  if ws > 1:                                # calibration and stream search differ per rank: start the timed section together
    dist.barrier(group=group)
    torch.cuda.synchronize(device)
  spin_done, coll_done = torch.cuda.Event(), torch.cuda.Event()
  with torch.cuda.stream(busy):
    torch.cuda._sleep(cycles)
    spin_done.record(busy)
  t0 = time.perf_counter()
  with torch.cuda.stream(issue):
    work = dist.all_gather_into_tensor(dst, src, group=group, async_op=True)
    work.wait()                             # stream-level: `issue` waits for the communicator's stream
    coll_done.record(issue)
  beside, t_coll = False, None
  while True:
    c, sdone = coll_done.query(), spin_done.query()
    if c and t_coll is None:
      t_coll = time.perf_counter() - t0
      beside = not sdone
    if sdone and c:
      break
    if time.perf_counter() - t0 > 30.0:
      raise RuntimeError('exchange-stream probe timed out')
  t_spin = time.perf_counter() - t0
  torch.cuda.synchronize(device)
  return beside, t_coll, t_spin


def check_exchange_stream(model_or_device, group=None):
  """The communicator's stream against the engine's launch stream and its side stream: {'beside_main', 'beside_side', 'ok'}
  (agreed over all ranks: `ok` is the minimum).  Call once after init_process_group, on every rank, before training."""
  from .executor import checked_side_stream
  if isinstance(model_or_device, torch.device):
    device = model_or_device
  else:
    device = next(model_or_device.parameters()).device
  main = torch.cuda.current_stream(device)
  side = checked_side_stream(device)
  bm, tm, sm = collective_runs_beside(main, device, group)
  bs, ts, ss = collective_runs_beside(side, device, group)
  flag = torch.tensor([1.0 if (bm and bs) else 0.0], device=device)
  dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
  return {'beside_main': bool(bm), 'beside_side': bool(bs), 'ok': bool(flag.item() > 0.5),
          'collective_ms': [round(1e3 * tm, 3), round(1e3 * ts, 3)], 'spin_ms': [round(1e3 * sm, 2), round(1e3 * ss, 2)]}


def _steer_stream_pool(device, main, side):
  """Leave torch's stream pool in a state where the NEXT stream it hands out runs beside both `main` and `side`.  The pool is a
  ring (32 streams per priority, handed out round robin) and HIP fixes a stream's hardware queue when it creates it, so:
  draw until a candidate overlaps with both, then draw (ring period - 1) more -- the next draw returns that very stream
  again.  ProcessGroupNCCL takes its stream from this ring when it creates a communicator.  (Re-creating the group and
  hoping does not work: every attempt advances the ring by a fixed number of draws -- the probe's own candidates, the
  communicator's, the burn -- and four draws land on the same hardware queue again: measured, six attempts, six times the
  side stream's queue.)  Returns True when a candidate was found."""
  from .executor import _overlap_ratio
  first = torch.cuda.Stream(device)
  period = None
  for i in range(1, 129):                   # ring period: draws until the first handle comes back
    if torch.cuda.Stream(device).cuda_stream == first.cuda_stream:
      period = i
      break
  if period is None:
    return False
  for _ in range(period):
    c = torch.cuda.Stream(device)
    if c.cuda_stream in (main.cuda_stream, side.cuda_stream):
      continue
    if _overlap_ratio(main, c) < 1.5 and _overlap_ratio(side, c) < 1.5:
      for _ in range(period - 1):
        torch.cuda.Stream(device)
      return True
  return False


def init_with_overlapping_exchange(init_fn, device, tries=3):
  """init_fn() -> creates the default process group (dist.init_process_group(...), same call on every rank).  Steers torch's
  stream pool so that the communicator's stream lands on a hardware queue of its own (_steer_stream_pool), creates the group
  and probes it; while some rank still reports a shared queue the group is destroyed and the procedure repeated.  Returns
  the last report with the number of attempts.

  init_fn is called up to `tries` times: it must be able to rendezvous again after a destroy_process_group() -- with
  env:// / tcp:// and a fixed MASTER_PORT the second bind can fail while the first store's socket lingers, so give every call
  a fresh port (bench.py: free_port()) or a file:// store.  A fresh group also clears a poisoned exchange (reset_poison)."""
  from .executor import checked_side_stream
  report = None
  for attempt in range(1, tries + 1):
    steered = _steer_stream_pool(device, torch.cuda.current_stream(device), checked_side_stream(device))
    init_fn()
    reset_poison()                          # whatever was in flight belonged to a group that no longer exists
    report = check_exchange_stream(device)
    report['attempts'] = attempt
    report['pool_steered'] = bool(steered)
    if report['ok'] or attempt == tries:
      return report
    dist.destroy_process_group()
  return report


def sync_gradients(optimizer, params=None, bucket_mb=64.0):
  """Average gradients across ranks before clipping (no-op in a single process).

  Flat-backed parameters (the score network) go through ``allreduce_flat_`` -- or, when the step armed an
  :class:`OverlappedExchange`, only wait for the buckets that are already in flight; any other parameter list is
  flattened into one temporary bucket."""
  if not is_distributed():
    return
  _check_poison()
  flat = getattr(optimizer, '_flat', None)
  if flat is None and hasattr(optimizer, '_bind'):
    try:
      flat = optimizer._bind()
    except RuntimeError:
      flat = None
  if flat is not None:
    armed = _armed.pop(id(flat), None)
    if armed is not None:
      armed.finish()
    else:
      allreduce_flat_(flat.grad, flat.n_train, bucket_mb=bucket_mb)
    return
  if params is None:
    params = [p for g in optimizer.param_groups for p in g['params']]
  grads = [p.grad for p in params if p.grad is not None]
  if not grads:
    return
  buf = torch.cat([g.reshape(-1) for g in grads])
  allreduce_flat_(buf, buf.numel(), bucket_mb=bucket_mb)
  off = 0
  for g in grads:
    g.copy_(buf[off:off + g.numel()].view_as(g))
    off += g.numel()


def shard_batch(batch):
  """This rank's contiguous slice of a global batch (global batch must divide the world size)."""
  ws, r = world_size(), rank()
  if ws == 1:
    return batch
  assert batch.shape[0] % ws == 0, 'global batch must be divisible by the number of ranks'
  per = batch.shape[0] // ws
  return batch[r * per:(r + 1) * per]
