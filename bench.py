#!/usr/bin/env python
"""bench.py -- TwinGAN G+D training images/sec at the 256x256 final progressive stage on MI355X.

Contract: ``python bench.py --gpus N --steps K --warmup W``.  N>1: one rank per GPU over RCCL -- either the caller
launches the ranks (``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...``: WORLD_SIZE is in the
environment) or, called plainly, this script re-launches itself under torch.distributed.run with N ranks on
127.0.0.1 (the reference's single-process ``--num_clones=N``, deployment/model_deploy.py:186-239, as one process
per MI355X).  One "step" = one full G+D step = one generator/encoder apply + one
discriminator apply (n_critic = 2, image_generation.py:640-652) over one synthetic batch of
``--batch`` (source, target) pairs per GPU, inputs resident in HBM.  value = pairs/sec over all ranks.

Extra objects on the JSON line (rank 0, N = 1): ``roofline`` for the dominant kernel family (HIP
events around every launch in a separate instrumented pass of the same step) and ``cpu_baseline``
(the torch-CPU oracle timed on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

BF16_MFMA_PEAK_TFLOPS = 2500.0     # MI355X dense bf16 (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0              # HBM3E spec (6.3 TB/s achievable)
MAX_LINE_BYTES = 6144               # the ONE JSON line of the contract stays below this (tables go to a side file)
GFLOP_PER_PAIR_256 = 335.0         # conv+FC MACs*2 of one G+D step at 256x256


METRIC = 'training images/sec (G+D step) at 256\u00d7256 final stage, 1/2/4/8 MI355X'      # BASELINE.json's metric, verbatim


def parse():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=6)
  ap.add_argument('--warmup', type=int, default=2)
  ap.add_argument('--config', type=int, default=3, choices=[0, 1, 2, 3, 4],
                  help="BASELINE.json configs[i]: 0 = plain PGGAN trainer, 4x4 stage-0, batch 16 (the reference's CPU-runnable "
                       "plumbing case, image_generation.py), 1 = 64x64 batch 64, 2 = 128x128 batch 32, 3 = 256x256 batch 16 "
                       "per GPU (the metric's), 4 = 3 + self-attention at 64x64 + spectral-norm discriminators + loss scale 128")
  ap.add_argument('--batch', type=int, default=None, help='pairs per GPU (default: the config\'s)')
  ap.add_argument('--hw', type=int, default=None)
  ap.add_argument('--max-ch', type=int, default=256)
  ap.add_argument('--precision', default=None, choices=['bf16', 'fp16', 'fp32'],
                  help='activation storage; default bf16, fp16 for --config 4 (the dtype BASELINE.json names for it)')
  ap.add_argument('--repeats', type=int, default=3, help='timed regions of --steps steps each; the median is reported')
  ap.add_argument('--no-graph', action='store_true', help='eager launches instead of hipGraph replay')
  ap.add_argument('--no-roofline', action='store_true')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--cpu-batch', type=int, default=2)
  ap.add_argument('--cpu-threads', type=int, default=0, help='0 = min(host cores, 32)')
  ap.add_argument('--cpu-baseline-only', action='store_true', help=argparse.SUPPRESS)
  ap.add_argument('--overlap', default='auto', choices=['auto', 'on', 'off'],
                  help='segmented backward + overlapped gradient all-reduce (auto: when N > 1)')
  ap.add_argument('--reduce-always', action='store_true',
                  help='one GPU: bring up a one-rank RCCL group and issue the per-segment all-reduces anyway (a one-rank sum '
                       'is the identity) -- exercises the N > 1 schedule and its diagnostics on a 1-GPU box')
  ap.add_argument('--launch-check', action='store_true',
                  help='no kernels: bring up the N ranks, build the parameter store and time the per-segment gradient '
                       'all-reduce schedule of a step (gloo on CPU when no GPU is visible) -- tests the launcher')
  args = ap.parse_args()
  hw, batch = {0: (4, 16), 1: (64, 64), 2: (128, 32), 3: (256, 16), 4: (256, 16)}[args.config]
  if args.precision is None:
    args.precision = 'fp16' if args.config == 4 else 'bf16'
  args.hw = args.hw or hw
  args.batch = args.batch or batch
  return args


def _free_port():
  import socket
  sk = socket.socket()
  sk.bind(('127.0.0.1', 0))
  port = sk.getsockname()[1]
  sk.close()
  return port


def spawn_ranks(n):
  """``bench.py --gpus N`` without a launcher: re-run this command line as N ranks (torch.distributed.run, local
  rendezvous on 127.0.0.1); rank 0 prints the JSON line, the exit code is the job's."""
  import subprocess
  env = dict(os.environ)
  env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: RCCL needs it on this driver
  env.setdefault('OMP_NUM_THREADS', '8')
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr',
         '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
  return subprocess.call(cmd, env=env)


def launch_check(args, world, rank, device):
  """The data-parallel plumbing of a step without its kernels: per backward segment, the all-reduce of that segment's
  range of the flat gradient buffer (dp.GradReducer over params.grad_phase ranges), then the sum is checked."""
  from twingan_amd import Config
  from twingan_amd.dp import GradReducer
  from twingan_amd.params import ParamStore, declare_twingan
  cfg = Config(hw=args.hw, max_ch=args.max_ch, precision=args.precision)
  store = declare_twingan(ParamStore(device), cfg).build(0)
  red = GradReducer(world, None)
  t0 = time.perf_counter()
  for _ in range(args.steps):
    for grp in store.GROUPS:
      store.grad[grp].fill_(float(rank + 1))
      for ph in sorted(store.phase_bounds[grp]):
        lo, hi = store.phase_bounds[grp][ph]
        red.start(store.grad[grp][lo:hi], n_buckets=1)
      red.finish()
      want = world * (world + 1) / 2.0
      assert float(store.grad[grp].min()) == want and float(store.grad[grp].max()) == want, 'all-reduce sum is wrong'
  if device.type == 'cuda':
    torch.cuda.synchronize()
  dist.barrier() if world > 1 else None
  dt = time.perf_counter() - t0
  st = red.stats()
  return dict(ms_per_step=1e3 * dt / max(args.steps, 1), rccl_world=world,
              allreduce_bytes_per_step=st['allreduce_bytes'] // max(args.steps, 1),
              collectives_per_step=st['collectives'] // max(args.steps, 1),
              exposed_allreduce_ms=round(st['exposed_allreduce_ms'], 4),
              grad_bytes={g: 4 * store.grad[g].numel() for g in store.GROUPS},
              segments={g: {str(p): list(b) for p, b in store.phase_bounds[g].items()} for g in store.GROUPS})


def synthetic_batch(batch, hw, dtype, device, rank):
  """a_source ~ U[0,1) seed 1234 (CelebA-shaped), b_source ~ U[0,1) seed 4321 (Getchu-shaped)."""
  ga = torch.Generator().manual_seed(1234 + rank)
  gb = torch.Generator().manual_seed(4321 + rank)
  a = torch.rand(batch, hw, hw, 3, generator=ga).to(device).to(dtype).contiguous()
  b = torch.rand(batch, hw, hw, 3, generator=gb).to(device).to(dtype).contiguous()
  return a, b


def one_step(tr, a, b):
  tr.run(a, b)      # generator / encoder apply   (n_critic_counter % 2 == 0)
  tr.run(a, b)      # discriminator apply (GP alphas drawn on device)


def roofline_pass(tr, a, b, steps=2):
  """Re-runs the same step eagerly with HIP events around every kernel launch (on the launch stream) and aggregates the
  launches (1) per kernel FAMILY = base symbol of the kernel the dispatch selected (all template instantiations of
  conv_tile_kernel are one family; entry-point name where one entry is one kernel pair, e.g. tg_norm_act_bwd) and (2) per
  exact layer shape.  Returns (roof, tables): `roof` is the compact object of the bench line -- the roofline of the
  family the step spends most time in: its algorithmic flops (bytes) / its summed launch time against the dense bf16
  MFMA (HBM) peak, whichever bounds it (intensity vs the 312 FLOP/B ridge) -- and `tables` everything else (per-shape
  rows, per-symbol rows, the PMC samples), written to a side file, never to the line."""
  from twingan_amd import _lib
  rec = []
  graph_mode, tr.use_graph = tr.use_graph, False        # per-launch events need eager launches
  one_step(tr, a, b)
  _lib.profiler = rec
  for _ in range(steps):
    one_step(tr, a, b)
  torch.cuda.synchronize()
  _lib.profiler = None
  tr.use_graph = graph_mode
  return summarize_launches([(name, tag, fl, by, e0.elapsed_time(e1), kname) for name, tag, fl, by, e0, e1, kname in rec], steps)


def summarize_launches(rec, steps, pmc=None):
  """(roof, tables) from launch records (entry point, shape tag, algorithmic flops, algorithmic bytes, ms, kernel symbol).
  Pure host code (tests/test_host_cpu.py feeds it a recorded pass and bounds the size of the line)."""
  ridge = 1e3 * BF16_MFMA_PEAK_TFLOPS / HBM_PEAK_GBS
  fam, syms, shapes = {}, {}, {}
  t_total = t_min_total = 0.0
  for name, tag, fl, by, ms, kname in rec:
    t_total += ms
    if fl or by:      # the launch's own lower bound at the two peaks
      t_min_total += max(fl / (BF16_MFMA_PEAK_TFLOPS * 1e12), by / (HBM_PEAK_GBS * 1e9)) * 1e3
    skey = kname if kname else name                      # the symbol rocprofv3 reports, else the entry point
    fkey = kernel_key(skey)[0]                           # family: base symbol without template arguments
    for key, table in ((fkey, fam), (skey, syms), ('%s[%s]' % (name, tag) if tag else name, shapes)):
      f = table.setdefault(key, dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
      f['ms'] += ms
      f['flops'] += fl
      f['bytes'] += by
      f['launches'] += 1
  t_total = max(t_total, 1e-9)

  def row(k, f):
    r = dict(kernel=k, share=round(f['ms'] / t_total, 4), launches=f['launches'] // steps,
             avg_us=round(1e3 * f['ms'] / f['launches'], 2),
             tflops=round(f['flops'] / (f['ms'] * 1e-3) / 1e12, 1) if f['ms'] else 0.0,
             gbs=round(f['bytes'] / (f['ms'] * 1e-3) / 1e9, 1) if f['ms'] else 0.0)
    # the family's fraction of ITS bound: MFMA peak above the ridge, HBM peak below it
    if f['flops'] and f['flops'] / max(f['bytes'], 1.0) > ridge:
      r['frac'] = round(r['tflops'] / BF16_MFMA_PEAK_TFLOPS, 4)
      r['bound'] = 'mfma'
    else:
      r['frac'] = round(r['gbs'] / HBM_PEAK_GBS, 4)
      r['bound'] = 'hbm'
    return r

  top_shapes = sorted(((k, f) for k, f in shapes.items() if f['bytes']), key=lambda kv: -kv[1]['ms'])
  fams = sorted(((kk, ff) for kk, ff in fam.items() if ff['bytes']), key=lambda kv: -kv[1]['ms'])
  k, f = fams[0]
  head = row(k, f)
  intensity = f['flops'] / max(f['bytes'], 1.0)
  if head['bound'] == 'mfma':
    roof = dict(bound='mfma', achieved=head['tflops'], peak=BF16_MFMA_PEAK_TFLOPS, unit='TFLOP/s', frac=head['frac'], traffic=None)
  else:
    roof = dict(bound='hbm', achieved=head['gbs'], peak=HBM_PEAK_GBS, unit='GB/s', frac=head['frac'], traffic=None)
  roof['kernel'] = k
  roof['time_share'] = head['share']
  roof['launches_per_step'] = head['launches']
  roof['intensity_flop_per_byte'] = round(intensity, 1)
  roof['avg_launch_us'] = head['avg_us']
  roof['algorithmic_bytes_per_launch'] = int(f['bytes'] / f['launches'])
  roof['algorithmic_flops_per_launch'] = int(f['flops'] / f['launches'])
  tables = dict(steps=steps)
  # measured HBM traffic and MFMA utilisation from the PMC passes (tools/pmc_kernels.sh -> profiles/rNN_pmc.json):
  # PMC runs are per layer shape; `traffic` is the measured HBM bytes per launch of the largest sampled layer shape of the
  # headline family (compare with `traffic_shape_algorithmic_bytes`, the algorithmic bytes of that same launch)
  pmc_rows, pmc_src = pmc if pmc is not None else load_pmc()
  if pmc_rows:
    def samples_of(base):
      rows = [e for e in pmc_rows if kernel_key(e.get('kernel', ''))[0] == base]
      rows.sort(key=lambda e: -e['algorithmic_bytes_per_launch'])
      return [{kk: e[kk] for kk in ('kernel', 'shape', 'hbm_bytes_per_launch', 'algorithmic_bytes_per_launch', 'traffic_over_algorithmic',
                                    'mfma_util', 'mfma_flops_over_algorithmic') if kk in e} for e in rows]
    samples = samples_of(k)
    if samples:
      roof['traffic'] = samples[0].get('hbm_bytes_per_launch')
      roof['traffic_shape'] = samples[0].get('shape')
      roof['traffic_shape_algorithmic_bytes'] = samples[0].get('algorithmic_bytes_per_launch')
      roof['traffic_over_algorithmic'] = samples[0].get('traffic_over_algorithmic')
      utils = [(e['mfma_util'], e['algorithmic_bytes_per_launch']) for e in samples if 'mfma_util' in e]
      if utils:      # weighted by the samples' sizes (a proxy for their share of the family's time), not the best one
        roof['pmc_mfma_busy'] = round(sum(u * wgt for u, wgt in utils) / max(sum(wgt for _, wgt in utils), 1), 4)
      roof['traffic_source'] = pmc_src
    tables['pmc_source'] = ('%s: rocprofv3 --pmc, one counter set per pass; FETCH_SIZE x2 (gfx950); mfma_util = SQ_VALU_MFMA_BUSY_CYCLES '
                            '/ (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), per launch of the listed layer shape' % pmc_src)
    tables['pmc_by_family'] = {kk: samples_of(kk) for kk in fam if samples_of(kk)}
  # the north-star quantity: MFMA utilisation of the conv kernels, time-weighted over the step.  Per launch the MFMA pipe
  # is busy flops / 1024 cycles per SIMD (measured identity for the 32x32x16 / 16x16x32 bf16 instructions, DESIGN.md
  # section 5), so utilisation = algorithmic flops / (launch time x dense peak); padding MFMAs are not counted.
  conv3 = [(kk, ff) for kk, ff in fam.items() if ff['flops'] and kk.startswith('conv_')]
  if conv3:
    t3 = sum(ff['ms'] for _, ff in conv3)
    f3 = sum(ff['flops'] for _, ff in conv3)
    roof['conv_mfma_util_time_weighted'] = round(f3 / (t3 * 1e-3) / (BF16_MFMA_PEAK_TFLOPS * 1e12), 4)
    roof['conv_kernel_time_share'] = round(t3 / t_total, 4)
  roof['kernel_time_ms_per_step'] = round(t_total / steps, 3)      # eager per-launch events: exceeds ms_per_step (stream overlap)
  roof['launches_per_step_total'] = len(rec) // steps
  # whole step against the two peaks: sum over launches of max(flops/MFMA peak, bytes/HBM peak) / measured time
  roof['step_roofline_frac'] = round(t_min_total / t_total, 4)
  roof['families'] = [row(kk, ff) for kk, ff in fams[:6]]
  tables['families'] = [row(kk, ff) for kk, ff in sorted(fam.items(), key=lambda kv: -kv[1]['ms'])]
  tables['symbols'] = [row(kk, ff) for kk, ff in sorted(syms.items(), key=lambda kv: -kv[1]['ms'])]
  tables['shapes'] = [row(kk, ff) for kk, ff in top_shapes]
  return roof, tables


def write_tables(tables, config):
  """The per-shape / per-symbol / PMC tables of the roofline pass: to a side file (TG_BENCH_TABLES, else
  gpurun_out/bench_tables_c<config>.json when that directory can be made, else the temp dir), path returned."""
  import tempfile
  path = os.environ.get('TG_BENCH_TABLES')
  cands = [path] if path else [os.path.join(ROOT, 'gpurun_out', 'bench_tables_c%d.json' % config),
                               os.path.join(tempfile.gettempdir(), 'bench_tables_c%d.json' % config)]
  for p in cands:
    try:
      os.makedirs(os.path.dirname(p), exist_ok=True)
      with open(p, 'w') as fh:
        json.dump(tables, fh, indent=0)
      if os.environ.get('TG_DUMP_SHAPES'):      # older tooling: the per-shape rows alone
        with open(os.environ['TG_DUMP_SHAPES'], 'w') as fh:
          json.dump(tables['shapes'], fh, indent=0)
      return os.path.relpath(p, ROOT) if p.startswith(ROOT) else p
    except OSError:
      continue
  return None


def kernel_key(name):
  """(base symbol, numeric template arguments) of a kernel name in any spelling: the mangled symbol rocprofv3 reports
  (..conv_tile_wres_kernelILi3ELi16ELi32ELi1EE..), its demangled form, or what tg_last_kernel() notes
  ("conv_tile_kernel<3,32,64,2,upcat>").  Non-numeric arguments (bools, tags) are dropped."""
  import re
  m = re.search(r'(conv_\w+?_kernel|conv_\w+_mfma)', name)
  if not m:
    return name, ()
  rest = name[m.end():]
  if rest.startswith('I'):      # mangled template argument list
    nums = re.findall(r'Li(\d+)E', rest.split('EEv')[0])
  else:
    t = re.match(r'<([^>]*)>', rest)
    nums = [v.strip() for v in t.group(1).split(',') if v.strip().isdigit()] if t else []
  return m.group(1), tuple(int(v) for v in nums)


def same_kernel(a, b):
  (ba, ta), (bb, tb) = kernel_key(a), kernel_key(b)
  n = min(len(ta), len(tb))
  return ba == bb and ta[:n] == tb[:n]


def load_pmc():
  """The committed PMC table of the newest round (profiles/rNN_pmc*.json, tools/pmc_kernels.sh)."""
  import glob
  files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc*.json')))
  if not files:
    return [], None
  try:
    rows = json.load(open(files[-1])).get('kernels', [])
  except Exception:
    return [], None
  return rows, os.path.relpath(files[-1], ROOT)


def cpu_baseline(args):
  """Runs cpu_baseline_child in a subprocess with a hard time limit so the bench line is always printed."""
  import subprocess
  cmd = [sys.executable, os.path.abspath(__file__), '--cpu-baseline-only', '--hw', str(args.hw), '--max-ch',
         str(args.max_ch), '--cpu-batch', str(args.cpu_batch), '--cpu-threads', str(args.cpu_threads)]
  try:
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
    line = [l for l in out.stdout.splitlines() if l.startswith('{')]
    if line:
      return json.loads(line[-1])
    return dict(value=None, unit='images/sec', cores=0, kind='port', sample='failed: ' + out.stderr[-200:])
  except subprocess.TimeoutExpired:
    return dict(value=None, unit='images/sec', cores=0, kind='port', sample='timed out after 240 s')


def cpu_baseline_child(args):
  """torch-CPU oracle (kind 'port': the TF-1.8 reference cannot run here) on a bounded sample:
  one G+D step (efficient schedule) at the bench resolution with a reduced batch, after one untimed
  warm-up G+D step.  Threads are capped: intra-op parallelism of these small convs stops scaling
  long before the host's core count."""
  from oracle import torch_ref as R
  cores = args.cpu_threads or min(os.cpu_count() or 1, 32)
  torch.set_num_threads(cores)
  rcfg = R.Config(hw=args.hw, max_ch=args.max_ch)
  P = R.init_params(rcfg, seed=0)
  opt = R.AdamState(P, rcfg)
  g = torch.Generator().manual_seed(1234)
  bsz = args.cpu_batch
  s, t = torch.rand(bsz, args.hw, args.hw, 3, generator=g), torch.rand(bsz, args.hw, args.hw, 3, generator=g)
  al = torch.rand(bsz, 1, 1, 1, generator=g)
  def timed(literal, budget, max_reps):
    R.train_step(P, opt, s, t, rcfg, al, al, counter=0, literal_schedule=literal)      # untimed warm-up G+D step
    R.train_step(P, opt, s, t, rcfg, al, al, counter=1, literal_schedule=literal)
    reps, t0 = 0, time.time()
    while True:
      R.train_step(P, opt, s, t, rcfg, al, al, counter=0, literal_schedule=literal)
      R.train_step(P, opt, s, t, rcfg, al, al, counter=1, literal_schedule=literal)
      reps += 1
      dt = time.time() - t0
      if dt > budget or reps >= max_reps:
        return reps, dt
  reps, dt = timed(False, 10.0, 8)
  # ... and the reference's LITERAL schedule (image_generation.py:631-646: every session.run computes BOTH gradient sets
  # and applies one of them), which is what the TF-1.x graph executes -- BASELINE.md section 3 quotes both
  lreps, ldt = timed(True, 6.0, 3)
  print(json.dumps(dict(value=round(bsz * reps / dt, 4), unit='images/sec', cores=cores, host_cores=os.cpu_count(),
                        kind='port', sample='%d G+D step(s), batch %d at %dx%d, fp32 torch-CPU oracle (oracle/torch_ref.py), efficient '
                               'schedule, %d threads of %d host cores, %.1f s' % (reps, bsz, args.hw, args.hw, cores, os.cpu_count() or 0, dt),
                        literal_schedule=dict(value=round(bsz * lreps / ldt, 4), unit='images/sec',
                                              sample='%d G+D step(s) computing both gradient sets per run as the reference graph does '
                                                     '(image_generation.py:631-646), %.1f s' % (lreps, ldt)))))


def main():
  args = parse()
  if args.cpu_baseline_only:
    return cpu_baseline_child(args)
  if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
    sys.exit(spawn_ranks(args.gpus))
  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  have_gpu = torch.cuda.is_available()
  if not have_gpu and not args.launch_check:
    raise SystemExit('bench.py needs a GPU (the HIP kernels have no CPU fallback); --launch-check tests the launcher')
  backend = 'nccl' if have_gpu else 'gloo'
  if world > 1:
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if have_gpu:
      torch.cuda.set_device(local_rank)
      dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
    else:
      dist.init_process_group(backend, rank=rank, world_size=world)
  elif have_gpu:
    torch.cuda.set_device(0)
    if args.reduce_always:      # a one-rank RCCL group: the collectives of the N > 1 schedule on one GPU
      os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
      os.environ.setdefault('MASTER_PORT', str(_free_port()))
      os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
      dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
  device = torch.device('cuda', local_rank if world > 1 else 0) if have_gpu else torch.device('cpu')
  observed_world = dist.get_world_size() if world > 1 else 1

  def base_line(value, ms_per_step, launch):
    return {
        'metric': METRIC if (args.hw == 256 and args.config == 3) else (
            'training images/sec (G+D step) at 256x256 + self-attention + spectral norm' if args.config == 4 else
            'training images/sec (G+D step), plain PGGAN trainer at 4x4 stage-0' if args.config == 0 else
            'training images/sec (G+D step) at %dx%d' % (args.hw, args.hw)),
        'value': value, 'unit': 'images/sec', 'n_gpus': observed_world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': args.precision, 'data': 'synthetic',
        'config': {'workload': ('plain PGGAN trainer (image_generation.py), 4x4 stage-0, batch %d per GPU (configs[0]): latent-noise '
                                'generator + one discriminator, max_ch %d, WGAN-GP, Adam; 1 step = G apply + D apply' % (
                                    args.batch, args.max_ch)) if args.config == 0 else
                               'TwinGAN %dx%d stage, batch %d per GPU (configs[%d]): E/G/2xD max_ch %d, UNet + per-domain '
                               'instance norm + pixel norm%s, WGAN-GP, Adam; 1 step = G apply + D apply' % (
                                   args.hw, args.hw, args.batch, args.config, args.max_ch,
                                   ' + self-attention at 64x64 + spectral-norm D + loss scale 128' if args.config == 4 else ''),
                   'global_batch': args.batch * observed_world, 'batch_per_gpu': args.batch,
                   'parallelism': 'dp%d' % observed_world, 'launch': launch,
                   'collective': ('%s all-reduce, world %d' % ('RCCL' if backend == 'nccl' else backend, observed_world))
                   if world > 1 else None,
                   'gflop_per_pair_model': GFLOP_PER_PAIR_256 if args.hw == 256 else None},
    }

  if args.launch_check:
    info = launch_check(args, world, rank, device)
    out = base_line(None, round(info['ms_per_step'], 3), 'launch-check (no kernels)')
    out['launch_check'] = info
    if rank == 0:
      print(json.dumps(out))
    if world > 1:
      dist.destroy_process_group()
    return

  # a 16-bit conv that the MFMA kernels do not take must fail here, not run 100 x slower under an MFMA label (ops._slow_dispatch)
  os.environ.setdefault('TG_STRICT_DISPATCH', '1')
  from twingan_amd import Config
  from twingan_amd.twingan import Trainer
  extra = {}
  if args.config == 4:
    # configs[4]: SAGAN attention (libs/self_attention.py) in E / G / D at 64x64, spectral-norm discriminators
    # (libs/sn.py), fp16 storage (TG_F16, --precision fp16 by default for this config) with the reference's static loss
    # scale 128 (model_inheritor.py:568-570); fp32 accumulation and master weights as on the bf16 path.
    extra = dict(do_self_attention=True, self_attention_hw=64, spectral_norm=True, loss_scale=128.0)
  cfg = Config(hw=args.hw, max_ch=args.max_ch, precision=args.precision, **extra)
  # auto: segmented when N > 1 -- and under --reduce-always, which exists to run the N > 1 schedule on one GPU
  overlap = (True if args.reduce_always else None) if args.overlap == 'auto' else args.overlap == 'on'
  if args.config == 0:      # configs[0]: the plain PGGAN trainer (generator from latent noise, one discriminator)
    from twingan_amd.image_generation import PgganTrainer
    tr = PgganTrainer(cfg, device=device, seed=0, world_size=world, use_graph=not args.no_graph, overlap=overlap)
  else:
    tr = Trainer(cfg, device=device, seed=0, world_size=world, use_graph=not args.no_graph, overlap=overlap)
  if args.reduce_always:
    tr.reducer.always = True
  dtype = {'bf16': torch.bfloat16, 'fp16': torch.float16, 'fp32': torch.float32}[args.precision]
  a, b = synthetic_batch(args.batch, args.hw, dtype, device, rank)

  for _ in range(max(args.warmup, 1)):      # at least one: the first graph-mode step is the capture
    one_step(tr, a, b)
  if not args.no_graph and not tr.use_graph:
    # the timed region must be the hipGraph replay the line says it is: never time a silent eager fallback
    sys.stderr.write('bench.py: hipGraph capture failed on rank %d (%s)\n' % (rank, tr.graph_fallback_reason))
    sys.exit(3)
  # the synthetic batch lives where the captured graphs read it (a loader would write its batches there): inputs resident
  # in HBM when the timed region starts, no per-run copy of the same bytes
  static = tr.static_inputs() if hasattr(tr, 'static_inputs') else None
  if static is not None and static[0] is not None:
    static[0].copy_(a)
    static[1].copy_(b)
    a, b = static
  torch.cuda.synchronize()
  tr.reducer.reset_stats()

  def timed_region():
    """EXACTLY --steps steps between barrier + synchronize on both sides; the MAX over ranks."""
    if world > 1:
      dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
      one_step(tr, a, b)
    torch.cuda.synchronize()
    if world > 1:
      dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
      tt = torch.tensor([dt], dtype=torch.float64, device=device)
      dist.all_reduce(tt, op=dist.ReduceOp.MAX)
      dt = float(tt.item())
    return dt

  # the pool's boxes (and one box over time) spread by several per cent: the region is timed --repeats times back to back
  # and the line reports the MEDIAN region (value, ms_per_step), with the fastest / slowest beside it
  regions = sorted(timed_region() for _ in range(max(args.repeats, 1)))
  elapsed = regions[len(regions) // 2]

  ms_per_step = 1e3 * elapsed / args.steps
  value = args.batch * world * args.steps / elapsed
  out = base_line(round(value, 3), round(ms_per_step, 3), 'hipGraph replay' if tr.use_graph else 'eager')
  out['timed_regions'] = dict(repeats=len(regions), steps_each=args.steps, reported='median',
                              value_max=round(args.batch * world * args.steps / regions[0], 3),
                              value_min=round(args.batch * world * args.steps / regions[-1], 3))
  out['config']['inputs'] = 'resident in the captured graphs\' static HBM buffers (no per-step host or device copy)'
  out['config']['backward_segments'] = {g: tr._nseg(g) for g in ('g', 'd')}
  if tr.capture_note:
    out['config']['capture_note'] = tr.capture_note
  if tr.reducer.active:
    # what the first run on real xGMI needs to be read: who took part, how the backward was cut, how many bytes went
    # through the collective per step and how long the compute stream waited for it (the exposed tail of the overlap)
    st = tr.reducer.stats()
    out['allreduce'] = dict(rccl_world=observed_world, backend=dist.get_backend() if dist.is_initialized() else None,
                            overlap='segmented' if tr.split else 'after the whole backward',
                            allreduce_bytes_per_step=st['allreduce_bytes'] // (args.steps * len(regions)),
                            collectives_per_step=st['collectives'] // (args.steps * len(regions)),
                            exposed_allreduce_ms_per_step=round(st['exposed_allreduce_ms'] * st['finishes'] / (args.steps * len(regions)), 4),
                            exposed_max_ms=round(st['exposed_max_ms'], 4), rank=rank)
  if args.hw == 256:
    tf = value * GFLOP_PER_PAIR_256 / 1e3 / world
    out['step_mfma_frac'] = round(tf / BF16_MFMA_PEAK_TFLOPS, 4)       # whole-step fraction of the conv roofline
  if rank == 0 and world == 1:
    if not args.no_roofline:
      out['roofline'], tables = roofline_pass(tr, a, b)
      out['roofline']['tables'] = write_tables(tables, args.config)
    if not args.no_cpu_baseline and args.config != 0:
      tr.close()
      del tr
      torch.cuda.empty_cache()
      out['cpu_baseline'] = cpu_baseline(args)
  if rank == 0:
    line = json.dumps(out)
    if len(line) > MAX_LINE_BYTES:      # the driver's parser lost round 5's 20 KB line: never again
      out['roofline'] = {k: v for k, v in out.get('roofline', {}).items() if not isinstance(v, (list, dict))}
      line = json.dumps(out)
    sys.stdout.flush()
    print(line, flush=True)
  if dist.is_initialized():
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
