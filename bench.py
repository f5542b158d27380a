#!/usr/bin/env python
"""Benchmark of the hot path: Soft-Truncation training step of the NCSN++/DDPM++ score network.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

One "step" = one call of ``losses.get_step_fn(...)(state, batch)`` -- time sampling, perturbation,
score-network forward and backward on the HIP engine, gradient all-reduce (N > 1), clip + Adam, EMA --
on one synthetic batch already resident in HBM.  Workload at N = 1 (BASELINE.json configs[1]):
DDPM++ (VP) CIFAR-10 32x32, per-GPU batch 128, fp32.  Scaling is weak: the per-GPU batch is fixed and
the global batch is 128 N.  Rank 0 prints ONE JSON line.

`python bench.py --gpus N` outside a launcher starts the N ranks itself (torch.distributed.run on 127.0.0.1).

The stdout line (short_line) is SHORT by construction -- below 4000 characters, asserted in tests/test_bench_line.py for N = 1 and
N = 8: the driver keeps the last 8000 characters of stdout, and the round-5 line (20 KB of per-kernel tables) arrived headless.  It
carries the contract's scalars and `config`, plus:
  roofline       the contraction kernel with the LARGEST TOTAL TIME in the step (whatever stream it runs on), timed with HIP events
                 on the launch stream in `--prof-steps` eager one-stream steps right after the timed ones:
                 achieved = algorithmic FLOPs per launch / average launch duration, against the ceiling of the matrix pipe for the
                 arithmetic that kernel runs: 2500 / 3 = 833 TFLOP/s fp32-equivalent for the fp16 two-way-split kernels (three fp16
                 MFMAs per fp32 product; labels .x2 / .x2p...), 157.3 for the f32-input MFMA kernels (MI355X_MICROARCH.md).
                 shares_chip = true: the kernel is a 3x3 weight gradient, which the timed steps run on the side stream beside the main
                 chain of the backward on one workgroup per CU; the bracket measures it with that same geometry.
                 traffic = HBM bytes per launch from the committed PMC summary of this workload (profiles/rNN[_workload]_traffic.json).
  roofline_best  the kernel with the highest fraction of its ceiling among those that take >= 2 % of the step.
  step_roofline  the whole step: BASELINE.md train FLOPs / image x images/s against 833, and the SURVEY 8(d) HBM model against 8 TB/s.
  cpu_baseline   the oracle's PyTorch-CPU restatement of the same training step (RefNet + torch Adam + EMA), timed on this box's host
                 cores on a bounded sample (five batch-8 steps + one batch-128 step; rank 0, N = 1 only).
  headline       one (images/s, ms/step, step fraction, dominant kernel + fraction + traffic, CPU images/s) tuple per workload measured in
                 the run (N = 1, default workload: BASELINE configs[2] / configs[4] nets too; celebahq256 adds its PC sampler).
  exchange_stream / exchange_serialised   N > 1: whether the communicator's stream runs beside the engine's launch and side streams on
                 every rank (engine/ddp.py: check_exchange_stream); exchange_serialised = true means a flat scaling curve is the queue.
  parity_probe   the benched build checks itself: per-sample losses of one batch-8 step against the oracle RefNet (N = 1 only).
  detail         path of the side file (--detail, default gpurun_out/bench_detail.json) with EVERYTHING the run measured: per-kernel
                 tables of every workload, full roofline / sampler / cpu_baseline / exchange_proxy / arithmetic_check objects.
One label = one kernel symbol of the rocprofv3 summaries (KERNEL_SYMBOL below): .x2p = LDS-DMA GEMM on plane operands, .x2p.h16/.h32/.h64
= its halo-tile form on the 16/32/64-wide maps, .x2p.k = its K-split form on small maps, .x2p.w32/.w16/.w8/.w4 = the planes weight
gradient per map width (bracket = kernel + its slab reduce).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
  if _p not in sys.path:
    sys.path.insert(0, _p)

import numpy as np
import torch
import torch.distributed as dist

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md "Peak FP32 (matrix)": the f32-input MFMA kernels (t64 / t128)
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md dense bf16 MFMA peak
# The x2 kernels (forward, data gradient and weight gradient of the 3x3 and 1x1 layers) issue three fp16 MFMAs per fp32
# product (two-way split of power-of-two-scaled operands), the x3 kernels (kept for shapes the x2 ones do not take) six
# bf16 MFMAs (three-way split); fp16 and bf16 MFMAs run at the same rate: the fp32-equivalent ceiling of the matrix pipe
# is the 16-bit peak / 3 resp. / 6.
PEAK_X3_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0
PEAK_X2_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 3.0


# rocprofv3 symbol of the kernel behind a profiler label (for the committed PMC summary, see traffic_of)
KERNEL_SYMBOL = {
  'conv3x3.wgrad.x2': 'x2::wgrad3_kernel<false>',
  'conv3x3.fwd.x2': 'x2::gemm_kernel<x2::ActLoader<false, 9>, EpFwd, 2>',
  'conv3x3.dgrad.x2': 'x2::gemm_kernel<x2::ActLoader<false, 9>, EpDgrad, 2>',
  'conv1x1.fwd.x2': 'x2::gemm_kernel<x2::ActLoader<true, 1>, EpFwd, 2>',
  'conv1x1.dgrad.x2': 'x2::gemm_kernel<x2::ActLoader<false, 1>, EpDgrad, 2>',
  'conv1x1.wgrad.x2': 'x2::wgemm_kernel<x2::RowsU<false, false>, x2::RowsU<true, true>, EpWgrad, true>',
  # plane operands (round 2): LDS-DMA staged forward / data gradient ('.k' = the K-split form of the small maps: EpSlab
  # partial tiles + a slab-sum launch), transpose-read weight gradient (one symbol per map width)
  'conv3x3.fwd.x2p': 'x2d::gemm_kernel<9, 128, EpFwd, 1, 0>',
  'conv3x3.dgrad.x2p': 'x2d::gemm_kernel<9, 128, EpDgrad, 1, 0>',
  'conv3x3.fwd.x2p.k': 'x2d::gemm_kernel<9, 128, EpSlab, 1, 0>',
  'conv3x3.dgrad.x2p.k': 'x2d::gemm_kernel<9, 128, EpSlab, 1, 0>',
  'conv1x1.fwd.x2p': 'x2d::gemm_kernel<1, 128, EpFwd, 1, 0>',
  'conv1x1.dgrad.x2p': 'x2d::gemm_kernel<1, 128, EpDgrad, 1, 0>',      # round 5: shortcut convolutions read their peer's dy planes
  'conv1x1.dgrad.x2p.k': 'x2d::gemm_kernel<1, 128, EpSlab, 1, 0>',
  # round 3: the halo-tile GEMM (one staged halo tile of the activations per channel group serves the nine taps)
  'conv3x3.fwd.x2p.h16': 'x2d::gemm_halo_kernel<16, EpFwd, 1>', 'conv3x3.dgrad.x2p.h16': 'x2d::gemm_halo_kernel<16, EpDgrad, 1>',
  'conv3x3.fwd.x2p.h32': 'x2d::gemm_halo_kernel<32, EpFwd, 1>', 'conv3x3.dgrad.x2p.h32': 'x2d::gemm_halo_kernel<32, EpDgrad, 1>',
  'conv3x3.fwd.x2p.h64': 'x2d::gemm_halo_kernel<64, EpFwd, 1>', 'conv3x3.dgrad.x2p.h64': 'x2d::gemm_halo_kernel<64, EpDgrad, 1>',
  'conv3x3.wgrad.x2p.w32': 'x2w::wgrad_kernel<32, 1>',
  'conv3x3.wgrad.x2p.w16': 'x2w::wgrad_kernel<16, 2>',     # round 4: eight-wave (two-group) workgroups on the 8- / 16-wide maps
  'conv3x3.wgrad.x2p.w8': 'x2w::wgrad_kernel<8, 2>',
  'conv3x3.wgrad.x2p.w4': 'x2w::wgrad_kernel<4, 1>',
}


def traffic_of(kind, path=None):
  """HBM-side bytes per launch of the kernel behind `kind`, from a committed PMC summary (default: the newest
  profiles/rNN_traffic.json, written by tools/profile_round.sh + tools/profile_summary.py: separate rocprofv3
  --pmc FETCH_SIZE / WRITE_SIZE passes of this same command, FETCH_SIZE doubled per the gfx950 note of
  MI355X_MICROARCH.md).  PMC counters cannot be read from inside this process, hence the file."""
  import glob
  if path is None:
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_traffic.json')))
    path = files[-1] if files else None
  sym = KERNEL_SYMBOL.get(kind)
  if path is None or sym is None:
    return None, None
  rec = json.load(open(path)).get(sym)
  if rec is None:
    return None, None
  return (rec['fetch_MB'] + rec['write_MB']) * 1e6, os.path.relpath(path, ROOT)


def kernel_peak(kind):
  if '.x2' in kind:                 # .x2 / .x2p / .x2p.k / .x2p.w32 ...
    return PEAK_X2_TFLOPS
  return PEAK_X3_TFLOPS if kind.endswith('.x3') else PEAK_F32_MFMA_TFLOPS
TRAIN_FLOPS_PER_IMG = {'cifar10_ddpmpp_nll_st': 65.072e9, 'imagenet32_ddpmpp_st': 65.072e9,
                       'celeba_uncsnpp_st': 252.128e9, 'celebahq_uncsnpp_st': 1598.169e9}

# forward activation traffic A_f of the fused model; a training step moves ~3 A_f per image
TRAIN_HBM_BYTES_PER_IMG = {'cifar10_ddpmpp_nll_st': 3 * 126.2e6, 'imagenet32_ddpmpp_st': 3 * 126.2e6,
                           'celeba_uncsnpp_st': 3 * 459.2e6, 'celebahq_uncsnpp_st': 3 * 3796.5e6}

WORKLOADS = {
  # name -> (config factory name, per-GPU batch, description)
  'cifar10': ('cifar10_ddpmpp_nll_st', 128, 'DDPM++ (VP) CIFAR-10 32x32, configs/vp/CIFAR10/ddpmpp_nll_st.py (BASELINE configs[1])'),
  'imagenet32': ('imagenet32_ddpmpp_st', 128, 'DDPM++ (VP) ImageNet32, configs/vp/IMAGENET32/ddpmpp_st.py (BASELINE configs[3])'),
  'celeba64': ('celeba_uncsnpp_st', 128, 'UNCSN++ (RVE) CelebA 64x64, configs/ve/CELEBA/uncsnpp_st.py (BASELINE configs[2])'),
  'celebahq256': ('celebahq_uncsnpp_st', 4, 'NCSN++ (VE) CelebA-HQ 256x256, configs/ve/celebahq/uncsnpp_st.py (BASELINE configs[4])'),
}


def parse():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=50)      # >= 50 timed steps after >= 10 warm-ups
  ap.add_argument('--warmup', type=int, default=10)
  ap.add_argument('--launch-check', action='store_true',
                  help='only bring up the N-rank process group (RCCL on GPUs, gloo without), count the ranks with one '
                       'all-reduce and print {"launch_check": true, "n_gpus": N}: tests the --gpus launcher without a GPU')
  ap.add_argument('--workload', default='cifar10', choices=sorted(WORKLOADS))
  ap.add_argument('--batch', type=int, default=0, help='per-GPU batch override')
  ap.add_argument('--fir', action='store_true', help='same net with model.fir=True (FIR resampling through upfirdn2d); SURVEY 8(d)')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-parity-probe', action='store_true')
  ap.add_argument('--no-kernel-timer', action='store_true')
  ap.add_argument('--prof-steps', type=int, default=3)
  ap.add_argument('--cpu-batch', type=int, default=8)
  ap.add_argument('--cpu-steps', type=int, default=5)
  ap.add_argument('--sampler-steps', type=int, default=6, help='PC-sampler iterations timed after the training steps (0 = skip)')
  ap.add_argument('--no-exchange-proxy', action='store_true',
                  help='N = 1 only: skip the second timed run with the gradient exchange forced on in a one-rank RCCL group')
  ap.add_argument('--no-extra-workloads', action='store_true',
                  help='N = 1, default workload only: skip the `workloads` object (BASELINE configs[2] / configs[4] nets, each with its '
                       'own step time, roofline, bounded CPU baseline, and the configs[4] PC sampler at N = 1000 / 2000)')
  ap.add_argument('--extra-steps', type=int, default=12, help='timed steps of each extra workload')
  ap.add_argument('--full-sampler-n', type=int, default=1000,
                  help='celebahq256 workload: one COMPLETE PC-sampler run on an N-point ladder (BASELINE configs[4] "1000-step"; 0 = skip)')
  ap.add_argument('--cpu-big-batch', type=int, default=128, help='batch of the single like-for-like CPU step (0 = skip)')
  ap.add_argument('--detail', default=os.path.join('gpurun_out', 'bench_detail.json'),
                  help='side file with everything the stdout line leaves out (per-kernel tables, workloads, sampler, probes)')
  ap.add_argument('--force-exchange', action='store_true',
                  help='N = 1: run the WHOLE benchmark (the reported value too) with the exchange forced on')
  return ap.parse_args()


def cpu_baseline(st, cfg_name, batch, steps, fir=False, big_batch=0, pc=True):
  """The oracle restatement (PyTorch CPU, oneDNN) of the same training step on the host cores."""
  import ref_torch
  cfg = st.configs.get_config(cfg_name)
  cfg.device = torch.device('cpu')
  if fir:
    cfg.model.fir = True
  sde = st.sde_lib.get_sde(cfg, None)
  # the product model only serves as the parameter initialiser here (same shapes / state_dict keys)
  torch.manual_seed(0)
  proto = st.models.ncsnpp.NCSNpp(cfg, sde)
  sd = {'module.' + k: v for k, v in proto.state_dict().items()}
  del proto
  ref = st.models.utils.DataParallel(ref_torch.RefNet(cfg, sd))
  opt = st.losses.get_optimizer(cfg, ref.parameters())
  ema = st.models.ema.ExponentialMovingAverage(ref.parameters(), decay=cfg.model.ema_rate)
  state = dict(optimizer=opt, model=ref, ema=ema, step=0)
  step_fn = st.losses.get_step_fn(cfg, sde, train=True, optimize_fn=st.losses.optimization_manager(cfg))
  x = st.datasets.synthetic_batch(cfg, batch, generator=torch.Generator().manual_seed(0))
  times = []
  for i in range(steps + 1):
    t0 = time.perf_counter()
    step_fn(state, x)
    times.append(time.perf_counter() - t0)
  med = float(np.median(times[1:]))
  out = {'value': batch / med, 'unit': 'images/s', 'cores': torch.get_num_threads(), 'kind': 'port',
         'sample': f'{cfg_name} full-size model, batch {batch}, median of {steps} training steps after 1 warm-up '
                   f'(oracle/ref_torch.RefNet + torch.optim.Adam + EMA, PyTorch CPU fp32)',
         'sec_per_step': med}
  # BASELINE.json configs[0] also names "1 PC sample step": one corrector + predictor iteration of the config's sampler
  # registry (euler_maruyama + none, the CPU-runnable pair), batch `batch`, median of `steps`
  try:
    if not pc:
      raise StopIteration
    ref.eval()
    xs = torch.randn(batch, cfg.data.num_channels, cfg.data.image_size, cfg.data.image_size)
    vec_t = torch.ones(batch) * 0.5
    pc = []
    with torch.no_grad():
      for i in range(steps + 1):
        t0 = time.perf_counter()
        xc, _ = st.sampling.shared_corrector_update_fn(xs, vec_t, sde, ref, st.sampling.NoneCorrector, True, 0.16, 1, cfg)
        st.sampling.shared_predictor_update_fn(xc, vec_t, sde, ref, st.sampling.EulerMaruyamaPredictor, False, True, cfg)
        pc.append(time.perf_counter() - t0)
    out['pc_iteration'] = {'sec': float(np.median(pc[1:])), 'batch': batch, 'image_evals_per_s': batch / float(np.median(pc[1:])),
                           'sample': 'euler_maruyama predictor + none corrector, one iteration'}
  except StopIteration:
    pass
  except Exception as e:
    out['pc_iteration'] = {'error': repr(e)[:200]}
  # a like-for-like batch (SURVEY.md 8(d) / BASELINE.md section 4 (ii): the benched per-GPU batch, 128): ONE step on the
  # warm state above -- about a minute of CPU work, the bounded sample of this leg
  if big_batch:
    try:
      ref.train()
      xb = st.datasets.synthetic_batch(cfg, big_batch, generator=torch.Generator().manual_seed(1))
      t0 = time.perf_counter()
      step_fn(state, xb)
      dt = time.perf_counter() - t0
      out[f'batch{big_batch}'] = {'value': big_batch / dt, 'unit': 'images/s', 'sec_per_step': dt, 'steps': 1}
    except Exception as e:
      out[f'batch{big_batch}'] = {'error': repr(e)[:200]}
  return out


def parity_probe(st, cfg_name, device):
  """The benched build checks itself (a build whose data gradients were computed from zero-filled weight
  blocks benchmarked fine): one batch-8 evaluation of the soft-truncation loss on the HIP engine and on the oracle
  RefNet with identical weights, batch, t_min and noise; per-sample losses and the network's parameter gradients
  (sum of squares) must agree.  Weights: the reference's own initialisation (seed 0)."""
  import copy
  import ref_torch
  from _model_util import patched_rng
  cfg = st.configs.get_config(cfg_name)
  cfg.model.dropout = 0.0                   # torch's CPU and the device draw different dropout masks
  cfg.device = device
  cfg_cpu = copy.deepcopy(cfg)
  cfg_cpu.device = torch.device('cpu')
  sde = st.sde_lib.get_sde(cfg, None)
  torch.manual_seed(0)
  net = st.models.ncsnpp.NCSNpp(cfg, sde).to(device)
  model = st.models.utils.DataParallel(net)
  net.engine().ensure_flat()
  sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
  ref = st.models.utils.DataParallel(ref_torch.RefNet(cfg_cpu, sd))
  batch = st.datasets.synthetic_batch(cfg_cpu, 8, generator=torch.Generator().manual_seed(7))
  out = {}
  res = []
  for m, c, b in ((model, cfg, batch.to(device)), (ref, cfg_cpu, batch)):
    loss_fn = st.losses.get_sde_loss_fn(c, sde, train=True)
    m.train()
    m.zero_grad()
    np.random.seed(3)
    with patched_rng(11):
      losses = loss_fn(m, b, importance_sampling=c.training.importance_sampling)
    torch.mean(losses).backward()
    gsq = sum(float((p.grad.double() ** 2).sum()) for p in m.parameters() if p.grad is not None)
    res.append((losses.detach().cpu().double(), gsq))
  (lp, gp), (lr, gr) = res
  out['loss_hip'] = [float(v) for v in lp]
  out['loss_ref'] = [float(v) for v in lr]
  out['loss_max_rel_err'] = float((lp - lr).abs().max() / lr.abs().max())
  out['grad_norm_rel_err'] = abs(gp ** 0.5 - gr ** 0.5) / max(gr ** 0.5, 1e-30)
  out['tolerance'] = 1e-3
  out['ok'] = bool(out['loss_max_rel_err'] <= 1e-3 and out['grad_norm_rel_err'] <= 1e-3)
  return out


def sampler_rate(st, cfg, sde, score_model, batch, steps, device):
  """Score evaluations per second of the config's own sampler: `steps` iterations of
  sampling.get_sampling_fn(...) on a batch of `batch` images (the reverse-time grid is shortened to `steps`
  points; every iteration runs the same corrector + predictor network evaluations as the full-length run)."""
  import copy
  sde_s = copy.copy(sde)
  sde_s.N = steps
  shape = (batch, cfg.data.num_channels, cfg.data.image_size, cfg.data.image_size)
  inverse_scaler = st.datasets.get_data_inverse_scaler(cfg)
  fn = st.sampling.get_sampling_fn(cfg, sde_s, shape, inverse_scaler, 1e-3 if cfg.training.sde == 'vpsde' else 1e-5)
  fn(score_model)                                   # warm-up (builds the inference program, captures its graph)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  _, nfe = fn(score_model)
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  evals = nfe + 1                                   # + the denoising evaluation at t = eps
  return {'score_evals_per_s': evals / dt, 'image_evals_per_s': evals * batch / dt, 'batch': batch, 'iterations': steps,
          'network_evals': evals, 'method': cfg.sampling.method, 'predictor': cfg.sampling.predictor,
          'corrector': cfg.sampling.corrector, 'ms_per_eval': 1e3 * dt / evals}


def traffic_file_for(workload):
  """The committed PMC summary of a workload's launches: profiles/rNN_traffic.json for the headline (CIFAR-10) shapes,
  profiles/rNN_<workload>_traffic.json for the others (tools/profile_round.sh / profile_workload.sh)."""
  import glob
  pat = 'r[0-9][0-9]_traffic.json' if workload in (None, 'cifar10', 'imagenet32') else f'r[0-9][0-9]_{workload}_traffic.json'
  files = sorted(glob.glob(os.path.join(ROOT, 'profiles', pat)))
  return files[-1] if files else None


def roofline_of(summ, prof_steps, ms_per_step, workload='cifar10'):
  """The `roofline` / `roofline_best` / `kernels` objects from a KernelTimer summary.

  roofline = the contraction kernel with the LARGEST TOTAL TIME in the step, whatever stream it runs on.  When that is one of the
  3x3 weight gradients ('.wgrad.x2p.*') the object carries shares_chip = true: in the timed steps they run on the side stream
  beside the main chain of the backward, on one workgroup per CU (csrc/conv_x2w.h); the event brackets behind `avg_us` are taken
  in one-stream eager steps with that same launch geometry, i.e. with half of every CU idle -- the figure is what the kernel does
  as launched, not what it could do with the chip to itself.  roofline_best = the contraction kernel with the highest fraction
  of its ceiling among those that take >= 2 % of the step.  traffic: the committed PMC summary of THIS workload's launches
  (traffic_file_for), null when none exists -- a kernel's bytes per launch depend on the shape."""
  shared = lambda k: '.wgrad.x2p' in k
  dom = max(summ, key=lambda k: summ[k]['total_ms'])
  tfile = traffic_file_for(workload)

  def obj(k):
    a = summ[k]
    traffic, traffic_src = traffic_of(k, tfile) if tfile else (None, None)
    return {'bound': 'mfma', 'achieved': a['tflops'], 'peak': kernel_peak(k), 'unit': 'TFLOP/s',
            'frac': a['tflops'] / kernel_peak(k), 'traffic': traffic, 'traffic_unit': 'bytes/launch',
            'traffic_source': traffic_src, 'kernel': k, 'symbol': KERNEL_SYMBOL.get(k),
            'peak_note': ('bf16 MFMA dense peak / 6 products per fp32 product' if k.endswith('.x3')
                          else 'fp16 MFMA dense peak / 3 products per fp32 product' if kernel_peak(k) == PEAK_X2_TFLOPS
                          else 'f32-input MFMA peak'),
            'avg_us': a['avg_us'], 'launches': a['count'], 'flops_per_launch': a['flops_per_launch'],
            'share_of_step': (a['total_ms'] / max(prof_steps, 1)) / ms_per_step, 'shares_chip': shared(k),
            'measured': f'{prof_steps} eager one-stream steps right after the timed steps, HIP events on the launch stream'}
  roof = obj(dom)
  roof['selection'] = 'largest total time among ALL contraction kernels of the step'
  big = [k for k in summ if (summ[k]['total_ms'] / max(prof_steps, 1)) / ms_per_step >= 0.02] or [dom]
  best = obj(max(big, key=lambda k: summ[k]['tflops'] / kernel_peak(k)))
  best['selection'] = 'highest fraction of its ceiling among the contraction kernels that take >= 2 % of the step'
  kernels = {k: {'tflops': round(v['tflops'], 2), 'frac_of_peak': round(v['tflops'] / kernel_peak(k), 3),
                 'avg_us': round(v['avg_us'], 1), 'launches': v['count'],
                 'total_ms_per_step': round(v['total_ms'] / max(prof_steps, 1), 3),
                 **({'shares_chip': True} if shared(k) else {})} for k, v in summ.items()}
  return roof, best, kernels


def extra_workload(st, name, device, args):
  """One more BASELINE workload inside the same run (N = 1): the same measurement as the headline -- K timed step_fn calls
  between device syncs after priming + warm-up, the dominant kernel's roofline from event-bracketed eager steps, the step
  against the matrix-pipe and HBM ceilings, a bounded CPU baseline of the same net -- as a sub-object of the JSON line.
  celebahq256 adds the config's PC sampler (reverse_diffusion + langevin) at N = 1000 and N = 2000 (the
  reference runs 2000, BASELINE.json says "1000-step"): the time grid has N points, a bounded number of iterations is timed."""
  import copy
  from importlib import import_module
  cfg_name, per_gpu_batch, desc = WORKLOADS[name]
  cfg = st.configs.get_config(cfg_name)
  cfg.device = device
  sde = st.sde_lib.get_sde(cfg, None)
  torch.manual_seed(0)
  score_model = st.models.utils.create_model(cfg, sde)
  eng = score_model.module.engine()
  eng.ensure_flat()
  optimizer = st.losses.get_optimizer(cfg, score_model.parameters())
  ema = st.models.ema.ExponentialMovingAverage(score_model.parameters(), decay=cfg.model.ema_rate)
  state = dict(optimizer=optimizer, model=score_model, ema=ema, step=0)
  step_fn = st.losses.get_step_fn(cfg, sde, train=True, optimize_fn=st.losses.optimization_manager(cfg))
  batch = st.datasets.synthetic_batch(cfg, per_gpu_batch, device=device, generator=torch.Generator().manual_seed(4321))
  steps = args.extra_steps
  for _ in range(2 + 4):
    step_fn(state, batch)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(steps):
    losses_ = step_fn(state, batch)
  torch.cuda.synchronize()
  elapsed = time.perf_counter() - t0
  ms = 1e3 * elapsed / steps
  ips = per_gpu_batch * steps / elapsed
  out = {'metric': f'training images/sec ({name})', 'value': ips, 'unit': 'images/s', 'ms_per_step': ms, 'steps': steps, 'warmup': 4,
         'dtype': 'f32', 'data': 'synthetic',
         'config': {'workload': desc, 'per_gpu_batch': per_gpu_batch, 'loss_mean': float(losses_.mean())}}
  tf = TRAIN_FLOPS_PER_IMG[cfg_name] * ips / 1e12
  gbs = (TRAIN_HBM_BYTES_PER_IMG[cfg_name] * per_gpu_batch + 16.0 * 4 * eng.flat.data.numel()) * (steps / elapsed) / 1e9
  out['step_roofline'] = {'bound': 'mfma', 'achieved': tf, 'peak': PEAK_X2_TFLOPS, 'unit': 'TFLOP/s', 'frac': tf / PEAK_X2_TFLOPS,
                          'frac_of_f32_input_mfma_peak': tf / PEAK_F32_MFMA_TFLOPS, 'hbm_algorithmic_GBps': gbs, 'frac_hbm': gbs / 8000.0}
  if not args.no_kernel_timer:
    timer = import_module('soft-truncation_amd.engine.profile').KernelTimer()
    eng.profiler = timer
    for _ in range(args.prof_steps):
      step_fn(state, batch)
    torch.cuda.synchronize()
    eng.profiler = None
    summ = timer.summary()
    if summ:
      out['roofline'], out['roofline_best'], out['kernels'] = roofline_of(summ, args.prof_steps, ms, workload=name)
  if name == 'celebahq256':
    out['sampler'] = {}
    sb = cfg.sampling.batch_size if hasattr(cfg.sampling, 'batch_size') else 16
    if args.full_sampler_n > 0:
      # BASELINE configs[4] "1000-step PC sampler throughput": ONE COMPLETE run of sampling.get_sampling_fn on an N-point
      # sigma ladder (model.num_scales = N), every iteration and the denoising step, measured -- not extrapolated
      try:
        out['sampler'][f'N{args.full_sampler_n}'] = pc_full_run(st, cfg, score_model, sb, args.full_sampler_n, device)
      except Exception as e:
        out['sampler'][f'N{args.full_sampler_n}'] = {'error': repr(e)[:200]}
    for n_grid in (2000,):
      try:
        out['sampler'][f'N{n_grid}'] = pc_rate(st, cfg, sde, score_model, sb, 12, device)
      except Exception as e:
        out['sampler'][f'N{n_grid}'] = {'error': repr(e)[:200]}
  del state, optimizer, ema, score_model, eng, step_fn
  torch.cuda.empty_cache()
  if not args.no_cpu_baseline:
    try:
      out['cpu_baseline'] = cpu_baseline(st, cfg_name, {'celeba64': 4, 'celebahq256': 1}.get(name, 2), 1, pc=False)
    except Exception as e:
      out['cpu_baseline'] = {'error': repr(e)[:200]}
  return out


def pc_full_run(st, cfg, score_model, batch, n_scales, device):
  """One complete run of the config's PC sampler (sampling.get_sampling_fn -> get_pc_sampler, sampling.py:365-433) with an
  `n_scales`-point noise ladder: N corrector + predictor iterations and the denoising step on a batch of `batch` images,
  timed between device syncs (the first call of the inference program -- eager run + hipGraph capture -- is made before)."""
  import copy
  cfg_n = copy.deepcopy(cfg)
  cfg_n.model.num_scales = n_scales
  sde_n = st.sde_lib.get_sde(cfg_n, None)
  shape = (batch, cfg.data.num_channels, cfg.data.image_size, cfg.data.image_size)
  eps = 1e-3 if cfg.training.sde == 'vpsde' else 1e-5
  inverse_scaler = st.datasets.get_data_inverse_scaler(cfg_n)
  warm = copy.copy(sde_n)
  warm.N = 2
  st.sampling.get_sampling_fn(cfg_n, warm, shape, inverse_scaler, eps)(score_model)     # builds / captures the inference program
  torch.cuda.synchronize()
  fn = st.sampling.get_sampling_fn(cfg_n, sde_n, shape, inverse_scaler, eps)
  t0 = time.perf_counter()
  x, nfe = fn(score_model)
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  evals = nfe + 1
  return {'measured': 'complete run', 'grid_points': n_scales, 'batch': batch, 'network_evals': evals, 'full_run_s': dt,
          'images_per_s_full_run': batch / dt, 'score_evals_per_s': evals / dt, 'image_evals_per_s': evals * batch / dt,
          'ms_per_eval': 1e3 * dt / evals, 'predictor': cfg.sampling.predictor, 'corrector': cfg.sampling.corrector,
          'finite': bool(torch.isfinite(x).all()), 'sample_min': float(x.min()), 'sample_max': float(x.max())}


def pc_rate(st, cfg, sde, score_model, batch, iterations, device):
  """Score evaluations per second of the PC sampler on the FULL-length time grid (sde.N points, timesteps = linspace(T, eps, N),
  sampling.py:404): `iterations` consecutive corrector + predictor iterations from the start of the grid, timed between
  syncs after two untimed ones (each iteration = 2 network evaluations for reverse_diffusion + langevin with n_steps = 1);
  the full run is N of these plus the denoising evaluation."""
  shape = (batch, cfg.data.num_channels, cfg.data.image_size, cfg.data.image_size)
  eps = 1e-3 if cfg.training.sde == 'vpsde' else 1e-5
  predictor = st.sampling.get_predictor(cfg.sampling.predictor.lower())
  corrector = st.sampling.get_corrector(cfg.sampling.corrector.lower())
  snr, n_steps = cfg.sampling.snr, cfg.sampling.n_steps_each
  score_model.eval()
  with torch.no_grad():
    x = sde.prior_sampling(shape).to(device)
    timesteps = torch.linspace(sde.T, eps, sde.N, device=device)

    def iterate(i):
      nonlocal x
      vec_t = torch.ones(shape[0], device=device) * timesteps[i]
      x, _ = st.sampling.shared_corrector_update_fn(x, vec_t, sde, score_model, corrector, cfg.training.continuous, snr, n_steps, cfg)
      x, _ = st.sampling.shared_predictor_update_fn(x, vec_t, sde, score_model, predictor, cfg.sampling.probability_flow,
                                                    cfg.training.continuous, cfg)
    for i in range(2):
      iterate(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(2, 2 + iterations):
      iterate(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
  evals = iterations * (n_steps + 1)
  full = sde.N * (n_steps + 1) + 1
  return {'score_evals_per_s': evals / dt, 'image_evals_per_s': evals * batch / dt, 'ms_per_eval': 1e3 * dt / evals, 'batch': batch,
          'grid_points': sde.N, 'iterations_timed': iterations, 'network_evals_timed': evals, 'network_evals_full_run': full,
          'full_run_s_extrapolated': full * dt / evals, 'images_per_s_full_run': batch / (full * dt / evals),
          'predictor': cfg.sampling.predictor, 'corrector': cfg.sampling.corrector, 'finite': bool(torch.isfinite(x).all())}


LINE_LIMIT = 4000     # characters of the stdout line (the driver keeps the last 8000 characters of stdout and parses the last line)


def _r(v, n=4):
  """Floats rounded to n significant-ish decimals for the stdout line (the detail file keeps full precision)."""
  if isinstance(v, float):
    return round(v, n) if abs(v) < 1e6 else float(f'{v:.6g}')
  return v


def _pick(d, keys, n=4):
  return {k: _r(d[k], n) for k in keys if isinstance(d, dict) and k in d}


def short_line(out, workload, detail_path=None):
  """The ONE stdout line: what the bench contract names and nothing else -- the contract's scalars, `config`, `roofline` (the
  contraction kernel with the largest total time in the step), `roofline_best`, `step_roofline`, `cpu_baseline`, `headline` (one
  triple per workload measured in the run), the exchange-stream verdict for N > 1, and `detail` = the path of the side file
  that holds everything else (per-kernel tables, every workload's full object, sampler legs, parity probe, arithmetic check,
  exchange proxy).  Stays below LINE_LIMIT characters by construction (tests/test_bench_line.py)."""
  def triple(o):
    t = {'images_per_s': round(o['value'], 1), 'ms_per_step': round(o['ms_per_step'], 3)}
    if 'step_roofline' in o:
      t['step_frac'] = round(o['step_roofline']['frac'], 4)
    if 'roofline' in o:
      t['dominant'] = o['roofline'].get('kernel')
      t['dominant_frac'] = round(o['roofline']['frac'], 4)
      if o['roofline'].get('traffic') is not None:
        t['dominant_traffic_MB'] = round(o['roofline']['traffic'] / 1e6, 1)
    if isinstance(o.get('cpu_baseline'), dict) and 'value' in o['cpu_baseline']:
      t['cpu_images_per_s'] = round(o['cpu_baseline']['value'], 3)
    return t
  head = {workload: triple(out)}
  for name, w in (out.get('workloads') or {}).items():
    if 'value' in w:
      head[name] = triple(w)
      for key, smp in (w.get('sampler') or {}).items():
        if isinstance(smp, dict) and 'ms_per_eval' in smp:
          head[name]['sampler_' + key] = _pick(smp, ('full_run_s', 'full_run_s_extrapolated', 'ms_per_eval', 'batch'), 3)
    else:
      head[name] = {'error': str(w.get('error', w))[:120]}
  first = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
           'dtype', 'data')
  line = {k: _r(out[k], 3) for k in first if k in out}
  line['config'] = _pick(out.get('config', {}), ('workload', 'per_gpu_batch', 'global_batch', 'parallelism', 'exchange'))
  if 'roofline' in out:
    line['roofline'] = _pick(out['roofline'], ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel', 'symbol', 'avg_us',
                                               'launches', 'flops_per_launch', 'share_of_step', 'shares_chip', 'traffic_source',
                                               'frac_of_library_gemm'))
  if 'roofline_best' in out:
    line['roofline_best'] = _pick(out['roofline_best'], ('kernel', 'achieved', 'frac', 'avg_us', 'share_of_step'))
  if 'step_roofline' in out:
    line['step_roofline'] = _pick(out['step_roofline'], ('bound', 'achieved', 'peak', 'unit', 'frac', 'frac_hbm',
                                                         'frac_of_f32_input_mfma_peak'))
  cb = out.get('cpu_baseline')
  if isinstance(cb, dict):
    line['cpu_baseline'] = _pick(cb, ('value', 'unit', 'cores', 'kind', 'sec_per_step', 'error'))
    if 'sample' in cb:
      line['cpu_baseline']['sample'] = cb['sample'][:160]
    for k, v in cb.items():
      if k.startswith('batch') and isinstance(v, dict):
        line['cpu_baseline'][k] = _pick(v, ('value', 'sec_per_step', 'error'))
  line['headline'] = head
  xs = out.get('exchange_stream')
  if isinstance(xs, dict):
    line['exchange_stream'] = _pick(xs, ('ok', 'beside_main', 'beside_side', 'pool_steered', 'attempts'))
    if 'error' in xs:
      line['exchange_stream']['error'] = str(xs['error'])[:120]
    if out.get('n_gpus', 1) > 1 and not xs.get('ok', False):
      line['exchange_serialised'] = True        # a flat scaling curve is then the communicator's queue, not the wire
  if isinstance(out.get('exchange_proxy'), dict) and 'delta_ms' in out['exchange_proxy']:
    line['exchange_proxy_delta_ms'] = round(out['exchange_proxy']['delta_ms'], 3)
  if isinstance(out.get('parity_probe'), dict):
    line['parity_probe'] = _pick(out['parity_probe'], ('ok', 'loss_max_rel_err', 'grad_norm_rel_err', 'tolerance'), 10)
    if 'error' in out['parity_probe']:
      line['parity_probe']['error'] = str(out['parity_probe']['error'])[:120]
  if isinstance(out.get('sampler'), dict) and 'ms_per_eval' in out['sampler']:
    line['sampler'] = _pick(out['sampler'], ('method', 'batch', 'ms_per_eval', 'image_evals_per_s'), 3)
  line['detail'] = detail_path
  text = json.dumps(line)
  if len(text) >= LINE_LIMIT:            # never reached with the objects above; a safety net, not a code path
    for k in ('sampler', 'parity_probe', 'roofline_best', 'exchange_proxy_delta_ms'):
      line.pop(k, None)
    text = json.dumps(line)
  assert len(text) < LINE_LIMIT, len(text)
  return text


def write_detail(out, path):
  """Everything the run measured, full precision, per-kernel tables included -- the file the stdout line names in `detail`."""
  try:
    os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
    with open(path, 'w') as f:
      json.dump(out, f, indent=1)
    return path
  except OSError as e:
    print(f'bench.py: could not write {path}: {e!r}', file=sys.stderr)
    return None


def arithmetic_check(device):
  """Accuracy of the two convolution paths on one layer of the workload's shape (128 -> 128, 3x3, 32x32, batch 24),
  each against float64: the fp16 two-way-split kernel (ws given) and the f32-input MFMA kernel (ws = NULL).
  Reported so that the `dtype: f32` claim of this line can be checked from the line itself."""
  from importlib import import_module
  lib = import_module('soft-truncation_amd.engine.lib').load()
  g = torch.Generator().manual_seed(0)
  N, C, H = 24, 128, 32
  x = torch.randn(N, C, H, H, generator=g).to(device)
  w = (torch.randn(C, C, 3, 3, generator=g) / 34.).to(device)
  ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
  shape = (C, 0, N, H, H, C, 3, 3, 1, 1)
  nbytes = int(lib.conv2d_fwd_ws_bytes(*shape))
  ws = torch.empty(nbytes // 4 + 64, dtype=torch.float32, device=device)
  stream = torch.cuda.current_stream(device).cuda_stream
  out = {}
  for name, wsp, wsb in (('fp16_two_way_split', ws.data_ptr(), nbytes), ('f32_input_mfma', None, 0)):
    y = torch.empty(N, C, H, H, dtype=torch.float32, device=device)
    lib.conv2d_fwd_f32(x.data_ptr(), C, None, 0, w.data_ptr(), 0, None, None, 0, None, 1.0, y.data_ptr(), N, H, H, C, H, H,
                       3, 3, 1, 1, wsp, wsb, stream)
    torch.cuda.synchronize()
    out[name + '_max_err_over_max_abs'] = float((y.double() - ref).abs().max() / ref.abs().max())
  out['split_path_selected'] = nbytes > 0
  return out


def pipe_probe(device):
  """What the 16-bit matrix pipe sustains on THIS box on random data: a 4096^3 fp16 library GEMM (torch.matmul ->
  hipBLASLt).  The nominal 2500 TFLOP/s is a 2.4 GHz figure; under matrix load the chip clocks to its power budget
  (MI355X_MICROARCH.md "DVFS give-back") and the vendor's own GEMM reaches about half of it.  Reported next to the
  roofline so that the fraction of the nominal peak can be read against what is attainable."""
  a = torch.randn(4096, 4096, device=device, dtype=torch.float16)
  b = torch.randn(4096, 4096, device=device, dtype=torch.float16)
  for _ in range(5):
    torch.matmul(a, b)
  torch.cuda.synchronize()
  s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  s.record()
  for _ in range(20):
    torch.matmul(a, b)
  e.record()
  torch.cuda.synchronize()
  return 2.0 * 4096 ** 3 * 20 / (s.elapsed_time(e) * 1e-3) / 1e12


def exchange_proxy(st, step_fn, state, batch, steps, device, init_group=True):
  """The fixed cost of the multi-GPU step on ONE GPU: the same step with the gradient exchange forced on in an RCCL
  process group of one rank -- backward replayed as one hipGraph per bucket segment, every 64 MB bucket of the flat gradient
  buffer all-reduced asynchronously on the communicator's stream as its last writer finishes, one wait and the division
  by the world size before clip / Adam (engine/ddp.py).  The all-reduce itself is the identity with one rank; what is
  measured is everything around it, which every rank of an N-GPU run pays on top of the wire time."""
  ddp = st.engine.ddp
  made = False
  stream_report = None
  if init_group and not dist.is_initialized():
    # a communicator whose stream shares a hardware queue with the launch or the side stream would serialise the exchange
    # behind the backward: probe it, and re-create the group until it runs beside both (engine/ddp.py)
    stream_report = ddp.init_with_overlapping_exchange(
      lambda: dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{free_port()}', rank=0, world_size=1,
                                      device_id=device), device)
    made = True
  calls = []
  real = dist.all_reduce

  def counting(*a, **k):
    calls.append(a[0].numel())
    return real(*a, **k)

  ddp.FORCE_SINGLE_RANK = True
  ddp.PROXY_TRAFFIC = True        # every bucket also crosses the communicator's stream as a real device copy (ddp.py)
  dist.all_reduce = counting
  try:
    for _ in range(4):                     # first forced step runs the segments eagerly, the second captures them
      step_fn(state, batch)
    torch.cuda.synchronize()
    del calls[:]
    t0 = time.perf_counter()
    for _ in range(steps):
      step_fn(state, batch)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
  finally:
    dist.all_reduce = real
    ddp.FORCE_SINGLE_RANK = False
    ddp.PROXY_TRAFFIC = False
    if made:
      dist.destroy_process_group()
  per_step = len(calls) // max(steps, 1)
  return {'ms_per_step': 1e3 * dt / steps, 'steps': steps, 'all_reduces_per_step': per_step,
          'bucket_MB': [round(4 * n / 2 ** 20, 1) for n in calls[:per_step]],
          'backend': 'nccl (RCCL), world_size 1', 'overlapped': bool(st.losses.OVERLAP_EXCHANGE),
          'traffic': 'every bucket also copied once through the communicator\'s stream (one-rank all-gather = device copy): '
                     'read + write of the 247 MB beside the backward',
          'exchange_stream': stream_report}


def free_port():
  import socket
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  return port


def relaunch(args):
  """`python bench.py --gpus N` outside a launcher: start the N ranks ourselves -- the same command the driver
  uses (torch.distributed.run, one process per GPU, rendezvous on 127.0.0.1) -- and pass their output through."""
  import subprocess
  env = dict(os.environ)
  env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: RCCL across processes needs it on this driver
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
         '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
  return subprocess.call(cmd, env=env)


def launch_check(world, rank, local_rank, out=sys.stdout):
  """The multi-process plumbing alone: process group up, every rank counted, rank 0 prints one JSON line."""
  on_gpu = torch.cuda.is_available()
  if on_gpu:
    torch.cuda.set_device(local_rank)
  device = torch.device('cuda', local_rank) if on_gpu else torch.device('cpu')
  n = 1
  backend = 'none'
  if world > 1:
    backend = 'nccl' if on_gpu else 'gloo'
    dist.init_process_group(backend, **({'device_id': device} if on_gpu else {}))
    t = torch.ones(1, device=device)
    dist.all_reduce(t)
    n = int(t.item())
    dist.barrier()
  if rank == 0:
    out.write(json.dumps({'launch_check': True, 'n_gpus': world, 'ranks_counted': n, 'backend': backend}) + '\n')
    out.flush()
  if world > 1:
    dist.destroy_process_group()


def quiet_stdout():
  """ONE JSON line on stdout: libraries print banners there (RCCL's version block at communicator creation, flushed at
  exit, i.e. after the line).  Point file descriptor 1 at stderr for the whole run and return a handle on the real one."""
  sys.stdout.flush()
  real = os.fdopen(os.dup(1), 'w')
  os.dup2(2, 1)
  return real


def device_of(local_rank):
  """This rank's GPU (a seam: tests/_bench_worker.py runs main() on CPU tensors with the checker library behind the engine)."""
  torch.cuda.set_device(local_rank)
  return torch.device('cuda', local_rank)


def init_group(device):
  dist.init_process_group('nccl', device_id=device)      # "nccl" is RCCL on ROCm
  dist.all_reduce(torch.zeros(1, device=device))         # first collective NOW: the communicator draws its stream from the steered pool


def device_sync():
  torch.cuda.synchronize()


def build_training(st, cfg, sde):
  """Random-init weights of the named architecture on cfg.device, optimizer, EMA, and the reference-shaped step function."""
  score_model = st.models.utils.create_model(cfg, sde)
  score_model.module.engine().ensure_flat()
  optimizer = st.losses.get_optimizer(cfg, score_model.parameters())
  ema = st.models.ema.ExponentialMovingAverage(score_model.parameters(), decay=cfg.model.ema_rate)
  state = dict(optimizer=optimizer, model=score_model, ema=ema, step=0)
  step_fn = st.losses.get_step_fn(cfg, sde, train=True, optimize_fn=st.losses.optimization_manager(cfg))
  return state, step_fn


def main():
  args = parse()
  if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
    sys.exit(relaunch(args))
  real_stdout = quiet_stdout()
  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if world > 1:
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
  if args.gpus != world and rank == 0:
    print(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; reporting n_gpus={world}', file=sys.stderr)
  if args.launch_check:
    return launch_check(world, rank, local_rank, real_stdout)
  device = device_of(local_rank)
  exchange_stream = None
  steered = None
  if world > 1:
    # steer torch's stream pool first, so that the communicator's stream lands on a hardware queue of its own (engine/ddp.py:
    # a communicator on the launch or the side stream's queue would serialise the overlapped exchange behind the backward)
    try:
      from importlib import import_module
      _ddp = import_module('soft-truncation_amd.engine.ddp')
      _ex = import_module('soft-truncation_amd.engine.executor')
      steered = bool(_ddp._steer_stream_pool(device, torch.cuda.current_stream(device), _ex.checked_side_stream(device)))
    except Exception as e:                                 # never fatal: the probe below reports what the communicator got
      print(f'bench.py: stream-pool steering failed: {e!r}'[:300], file=sys.stderr)
    init_group(device)

  import soft_truncation_amd as st
  st.sampling.PROGRESS = False                             # no progress bars: stdout carries one line, stderr stays short
  cfg_name, per_gpu_batch, desc = WORKLOADS[args.workload]
  if args.batch:
    per_gpu_batch = args.batch
  cfg = st.configs.get_config(cfg_name)
  cfg.device = device
  if args.fir:
    cfg.model.fir = True
    desc += ' with model.fir=True'
  st.engine.ddp.seed_everything(cfg.seed)                  # numpy shared (t_min), torch per rank

  if args.force_exchange and world == 1:
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{free_port()}', rank=0, world_size=1, device_id=device)
    st.engine.ddp.FORCE_SINGLE_RANK = True
  sde = st.sde_lib.get_sde(cfg, None)
  state, step_fn = build_training(st, cfg, sde)
  score_model = state['model']
  batch = st.datasets.synthetic_batch(cfg, per_gpu_batch, device=device,
                                      generator=torch.Generator().manual_seed(1234 + rank))
  if world > 1:
    # does the communicator's stream run beside the engine's launch and side streams on every rank?  (reported; a torchrun
    # rendezvous cannot be re-created from here, so a serialised exchange shows up in the line instead of being repaired)
    try:
      exchange_stream = st.engine.ddp.check_exchange_stream(device)
      exchange_stream['pool_steered'] = steered
    except Exception as e:
      exchange_stream = {'error': repr(e)[:200]}

  def sync():
    device_sync()
    if world > 1:
      dist.barrier()
      device_sync()

  # Two untimed priming steps before the W warm-up steps: the engine runs a context eagerly on first use and captures
  # its hipGraphs on the second, so the timed region never contains a capture whatever W is.
  for _ in range(2 + args.warmup):
    step_fn(state, batch)
  sync()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    losses_ = step_fn(state, batch)
  sync()
  elapsed = time.perf_counter() - t0
  proxy = None
  if world == 1 and not args.no_exchange_proxy and not args.force_exchange:
    try:
      proxy = exchange_proxy(st, step_fn, state, batch, args.steps, device)
      proxy['ms_per_step_without'] = 1e3 * elapsed / args.steps
      proxy['delta_ms'] = proxy['ms_per_step'] - proxy['ms_per_step_without']
    except Exception as e:                 # reported, never fatal
      proxy = {'error': repr(e)[:300]}
  # Kernel durations for the roofline object: the timed steps replay hipGraphs, and HIP events cannot be
  # recorded inside a replayed graph, so the same step is run `--prof-steps` more times right here with eager
  # launches, every contraction launch bracketed by events on the launch stream.
  timer = None
  if not args.no_kernel_timer and rank == 0:
    from importlib import import_module
    timer = import_module('soft-truncation_amd.engine.profile').KernelTimer()
    score_model.module.engine().profiler = timer
  if not args.no_kernel_timer:
    for _ in range(args.prof_steps):
      step_fn(state, batch)
    sync()
  score_model.module.engine().profiler = None
  if world > 1:
    t = torch.tensor([elapsed], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

  if rank == 0:
    global_batch = per_gpu_batch * world
    ips = global_batch * args.steps / elapsed
    out = {
      'metric': 'training images/sec (DDPM++ CIFAR-10 32x32)' if args.workload == 'cifar10'
                else f'training images/sec ({args.workload})',
      'value': ips, 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
      'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak',
      'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
      'arithmetic': 'fp32 tensors and fp32 accumulation everywhere; the large convolutions evaluate each fp32 product '
                    'from split operands on the 16-bit matrix pipe: power-of-two-scaled '
                    'operands as two fp16 terms, 3 MFMAs per product (forward, data gradient and weight gradient of the '
                    '3x3 and 1x1 layers) -- with errors at '
                    'the fp32 rounding level (arithmetic_check; parity-tested against the double-precision oracle at '
                    'the same tolerance as the f32-input MFMA path)',
      'config': {'workload': desc, 'per_gpu_batch': per_gpu_batch, 'global_batch': global_batch,
                 'parallelism': f'dp{world}', 'loss_mean': float(losses_.mean()),
                 'hipgraph_replays': score_model.module.engine().graph_replays},
    }
    if exchange_stream is not None:
      out['exchange_stream'] = exchange_stream
    if proxy is not None:
      out['exchange_proxy'] = proxy
    if args.force_exchange and world == 1:
      out['config']['exchange'] = 'forced on in a one-rank RCCL group (--force-exchange)'
    step_tflops = TRAIN_FLOPS_PER_IMG[cfg_name] * ips / world / 1e12
    out['step_roofline'] = {'bound': 'mfma', 'achieved': step_tflops, 'peak': PEAK_X2_TFLOPS,
                            'unit': 'TFLOP/s', 'frac': step_tflops / PEAK_X2_TFLOPS,
                            'frac_of_f32_input_mfma_peak': step_tflops / PEAK_F32_MFMA_TFLOPS,
                            'note': 'whole training step per GPU: BASELINE.md train FLOPs/img x images/s, against the '
                                    'matrix-pipe ceiling of the kernels that carry ~95 % of them (fp16 MFMA peak / 3 '
                                    'passes per fp32 product); SURVEY.md 8(d) frac_mfma against the 157.3 TFLOP/s of the '
                                    'f32-input MFMA is the second figure'}
    if timer is not None:
      summ = timer.summary()
      if summ:
        out['roofline'], out['roofline_best'], out['kernels'] = roofline_of(summ, args.prof_steps, 1e3 * elapsed / args.steps,
                                                                            workload=args.workload)
    hbm_bytes = TRAIN_HBM_BYTES_PER_IMG.get(cfg_name)
    if hbm_bytes is not None:
      gbs = (hbm_bytes * per_gpu_batch + 16.0 * 4 * score_model.module.engine().flat.data.numel()) * (args.steps / elapsed) / 1e9
      out['step_roofline'].update({'hbm_algorithmic_GBps': gbs, 'frac_hbm': gbs / 8000.0,
                                   'hbm_note': 'SURVEY.md 8(d) fused-traffic model: 3 x A_f bytes per image + 16 x 4 B per '
                                               'parameter per step, against 8 TB/s'})
    if world == 1:
      out['arithmetic_check'] = arithmetic_check(device)
      try:
        lib_tf = pipe_probe(device)
        if 'roofline' in out and out['roofline']['peak'] == PEAK_X2_TFLOPS:
          out['roofline'].update({
            'library_gemm_fp16_tflops': lib_tf,
            'frac_of_library_gemm': 3.0 * out['roofline']['achieved'] / lib_tf,
            'library_note': 'torch.matmul (hipBLASLt) fp16 4096^3 on random data, timed in this process: what the 16-bit '
                            'matrix pipe sustains at its power-limited clock; the kernel issues 3 MFMAs per fp32 product, so '
                            'frac_of_library_gemm = 3 x achieved / library rate'})
      except Exception as e:   # reported, never fatal
        out['pipe_probe_error'] = repr(e)
    if world == 1 and args.sampler_steps > 0:        # like the CPU baseline: only in the single-GPU run
      try:
        out['sampler'] = sampler_rate(st, cfg, sde, score_model, per_gpu_batch, args.sampler_steps, device)
      except Exception as e:                         # the reference's RVE sampling raises: report, do not fail
        out['sampler'] = {'error': repr(e)[:200]}
    if world == 1 and not args.no_cpu_baseline:
      out['cpu_baseline'] = cpu_baseline(st, cfg_name, args.cpu_batch, args.cpu_steps, args.fir, big_batch=args.cpu_big_batch)
    if world == 1 and not args.no_parity_probe:
      try:
        del state, score_model, step_fn              # free the benched replica's arenas before building the probe's
        torch.cuda.empty_cache()
        out['parity_probe'] = parity_probe(st, cfg_name, device)
      except Exception as e:
        out['parity_probe'] = {'error': repr(e)[:300]}
    if world == 1 and args.workload == 'cifar10' and not args.no_extra_workloads and not args.fir and not args.batch:
      # north_star: "throughput on 32x32 AND 256x256 batches": BASELINE configs[2] / configs[4] nets in the same driver-timed run
      try:
        del state, score_model, step_fn                  # (already gone when the parity probe ran)
      except NameError:
        pass
      torch.cuda.empty_cache()
      out['workloads'] = {}
      for name in ('celeba64', 'celebahq256'):
        try:
          out['workloads'][name] = extra_workload(st, name, device, args)
        except Exception as e:
          out['workloads'][name] = {'error': repr(e)[:300]}
    detail = write_detail(out, args.detail)
    real_stdout.write(short_line(out, args.workload, detail) + '\n')
    real_stdout.flush()
  if world > 1 or (args.force_exchange and dist.is_initialized()):
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
