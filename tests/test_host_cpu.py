"""CPU-side checks: the C-ABI library loads and exports every symbol include/twingan_hip.h declares,
the ctypes signatures cover the header, the host-side mirror (parameter schema, channel schedule,
padding rules, step schedule) agrees with the oracle, and the product path refuses to run without
a GPU (no CPU fallback)."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
  src = open(os.path.join(ROOT, 'include', 'twingan_hip.h')).read()
  src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
  return sorted(set(re.findall(r'\b(tg_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
  import ctypes
  from twingan_amd import _lib
  assert os.path.exists(_lib.LIB_PATH), 'build first: python -c "import __graft_entry__ as g; g.build()"'
  lib = ctypes.CDLL(_lib.LIB_PATH)
  syms = header_symbols()
  assert len(syms) >= 30
  for s in syms:
    assert hasattr(lib, s), 'missing export %s' % s


def test_ctypes_signatures_cover_header():
  from twingan_amd import _lib
  assert sorted(_lib.SIGNATURES) == header_symbols()
  lib = _lib.load()
  assert lib.tg_version() >= 100
  assert lib.tg_last_error() is not None


def test_conv_desc_layout_matches_header():
  import ctypes
  from twingan_amd._lib import TgConvDesc
  assert ctypes.sizeof(TgConvDesc) == 16 * 4      # 14 int32 + 1 float + groups, no padding


def test_invalid_descriptor_is_rejected_without_gpu():
  """Argument validation happens before any launch, so it can be exercised on a CPU-only box."""
  import ctypes
  from twingan_amd import _lib
  lib = _lib.load()
  d = _lib.TgConvDesc()
  d.n, d.hin, d.win, d.cin, d.hout, d.wout, d.cout = 1, 4, 4, 8, 9, 4, 8      # hout inconsistent with stride 1
  d.kh = d.kw = 3
  d.pad_t = d.pad_l = 1
  rc = lib.tg_conv2d_fwd(ctypes.byref(d), 16, 16, None, 16, None)
  assert rc == -1 and b'inconsistent' in lib.tg_last_error()
  d.hout = 4
  d.kh = 8                                    # direct kernels: up to 7x7 (to-RGB layers with the larger filter)
  assert lib.tg_conv2d_fwd(ctypes.byref(d), 16, 16, None, 16, None) == -1
  d.kh, d.algo = 5, 1                         # the MFMA kernels: 1x1, 3x3, dense 4x4 VALID
  assert lib.tg_conv2d_fwd(ctypes.byref(d), 16, 16, None, 16, None) == -1
  assert lib.tg_pointwise_conv_fwd(16, 16, None, 16, 10, 8, 8, 0, 0, 0.2, 0, None) == -4     # TG_ENOSUP


def test_pack_sizes():
  import ctypes
  from twingan_amd import _lib
  lib = _lib.load()
  d = _lib.TgConvDesc()
  d.n, d.hin, d.win, d.cin, d.hout, d.wout, d.cout = 16, 4, 4, 264, 4, 4, 256
  d.kh = d.kw = 3
  d.pad_t = d.pad_l = 1
  assert lib.tg_conv2d_pack_elems(ctypes.byref(d), 0) == 256 * 9 * 272          # rows pad 64, inner pad 16
  assert lib.tg_conv2d_pack_elems(ctypes.byref(d), 1) == 320 * 9 * 256
  d.cin, d.kh, d.kw, d.pad_t, d.pad_l, d.hout, d.wout = 256, 4, 4, 0, 0, 1, 1     # 4x4 VALID on 4x4 -> dense 4096
  assert lib.tg_conv2d_pack_elems(ctypes.byref(d), 0) == 256 * 4096
  assert lib.tg_conv2d_pack_elems(ctypes.byref(d), 1) == 4096 * 256


def test_no_cpu_fallback():
  from twingan_amd import ops
  from twingan_amd._lib import TgError
  with pytest.raises(TgError):
    ops.conv2d(torch.zeros(1, 4, 4, 8), torch.zeros(3, 3, 8, 8))
  with pytest.raises(TgError):
    ops.norm_act(torch.zeros(1, 4, 4, 8), torch.ones(8), torch.zeros(8))
  with pytest.raises(TgError):
    ops.abs_diff_mean(torch.zeros(4), torch.zeros(4))


def test_product_does_not_import_oracle():
  for dirpath, _, files in os.walk(os.path.join(ROOT, 'twingan_amd')):
    for f in files:
      if f.endswith('.py'):
        src = open(os.path.join(dirpath, f)).read()
        assert not re.search(r'^\s*(from|import)\s+oracle', src, flags=re.M), f


def test_channel_schedule_and_padding_rules():
  from twingan_amd.ops import ConvSpec
  from twingan_amd.params import get_num_channels, max_stage_of, mbstd_cpad
  assert [get_num_channels(s) for s in range(8)] == [256, 256, 256, 128, 64, 32, 16, 8]
  assert max_stage_of(4) == 0 and max_stage_of(256) == 6
  assert mbstd_cpad(256) == 264 and mbstd_cpad(8) == 16 and mbstd_cpad(32) == 40
  s = ConvSpec(3)
  assert (s.pad_t, s.out_hw(7, 5)) == (1, (7, 5))
  s = ConvSpec(4, 'VALID')
  assert (s.pad_t, s.out_hw(4, 4)) == (0, (1, 1))
  s = ConvSpec(4, 'SAME')
  assert s.pad_t == 1                               # TF SAME with even k: low side gets floor((k-1)/2)


@pytest.mark.parametrize('hw,max_ch,growing', [(256, 256, False), (64, 256, False), (16, 32, True), (4, 16, False)])
def test_param_schema_matches_oracle(hw, max_ch, growing):
  from oracle import torch_ref as R
  from twingan_amd import Config
  from twingan_amd.params import ParamStore, declare_twingan
  cfg = Config(hw=hw, max_ch=max_ch, is_growing=growing, alpha_grow=0.5)
  store = declare_twingan(ParamStore('cpu'), cfg).build(seed=0)
  ref = R.init_params(R.Config(hw=hw, max_ch=max_ch, is_growing=growing, alpha_grow=0.5))
  sd = store.state_dict()
  assert set(sd) == set(ref)
  for k in ref:
    assert tuple(sd[k].shape) == tuple(ref[k].shape), k
  assert set(store.names('g')) == set(R.generator_var_names(ref))
  assert set(store.names('d')) == set(R.discriminator_var_names(ref))
  if hw == 256:
    assert store.numel('g') == sum(ref[k].numel() for k in R.generator_var_names(ref))
    assert store.numel('d') == 2 * 4590097
  # init semantics: gamma 1, beta / biases 0, weights ~ N(0, 0.02), physical padding rows zero
  any_w = sd['generator/block_4x4x%d/Conv/weights' % min(256, max_ch)]
  assert 0.015 < float(any_w.std()) < 0.025
  tail = store['discriminator_s/before_fc_1x1x%d/Conv/weights' % max_ch]
  assert float(tail[:, :, max_ch + 1:, :].abs().max()) == 0.0
  # state-dict round trip with missing variables (pggan_runner.py:136-146 ignore_missing_vars)
  partial = {k: v for k, v in sd.items() if 'from_rgb' not in k}
  store.load_state_dict(partial, strict=False)
  with pytest.raises(KeyError):
    store.load_state_dict(partial, strict=True)
  # every variable's gradient is a view into the group's flat gradient buffer
  for k in store.names():
    g = store.specs[k]['group']
    base = store.grad[g]
    assert base.data_ptr() <= store[k].grad.data_ptr() < base.data_ptr() + base.numel() * 4


def test_variables_match_what_the_reference_creates():
  """tests/golden/variable_schema.json: the variables twingan.GanModel._clone_fn created when the reference's own code
  was executed (oracle/ref_runner.py, tools/make_golden.py) at full width -- names, shapes, trainable or not, and the
  mean / std of what its initialisers drew.  The product must declare exactly those."""
  import json
  import os
  from twingan_amd import Config
  from twingan_amd.params import ParamStore, declare_twingan
  from test_golden import PRODUCT_FIELD
  with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'variable_schema.json')) as fh:
    schema = json.load(fh)
  assert len(schema) >= 4
  for name, case in schema.items():
    cfg = Config(**{PRODUCT_FIELD.get(k, k): v for k, v in case['config'].items()})
    store = declare_twingan(ParamStore('cpu'), cfg).build(seed=0)
    ref = case['variables']
    train = {k: tuple(store.specs[k]['shape']) for k in store.specs}
    state = {k: tuple(v.shape) for k, v in store.state.items() if not k.startswith('renorm/')}      # device scalars
    assert train == {k: tuple(v['shape']) for k, v in ref.items() if v['trainable']}, name
    want_state = {k: tuple(v['shape']) or (1,) for k, v in ref.items() if not v['trainable']}
    assert {k: (v or (1,)) for k, v in state.items()} == want_state, name
    sd = store.state_dict(include_state=True)
    for k, v in ref.items():
      got = sd[k].double()
      if v['std'] == 0.0:                       # constant initialisers: zeros / ones
        assert float((got - v['mean']).abs().max()) == 0.0, (name, k)
      elif got.numel() >= 4096:                 # random initialisers: same distribution
        assert abs(float(got.std()) - v['std']) < 0.08 * v['std'], (name, k, float(got.std()), v['std'])
        assert abs(float(got.mean()) - v['mean']) < 0.1 * v['std'], (name, k)


def test_step_schedule_counters():
  """n_critic alternation and counters (image_generation.py:640-652) without touching the GPU."""
  from twingan_amd import Config
  from twingan_amd.twingan import Trainer
  tr = Trainer(Config(hw=8, max_ch=8), device='cpu')
  calls = []
  tr.g_step = lambda s, t: calls.append('g')
  tr.d_step = lambda s, t, a=None, b=None: calls.append('d')
  for _ in range(5):
    tr.run(None, None)
  # global_step advances at the end of the run that completes a cycle (see Trainer._advance_counters)
  assert calls == ['g', 'd', 'g', 'd', 'g'] and tr.global_step == 2 and tr.n_critic_counter == 5


def test_stage_schedule_matches_pggan_runner():
  """pggan_runner.py:90-109: stage names, growing/stable alternation, steps = images / batch, last stage open-ended."""
  from twingan_amd.runner import LAST_STAGE_STEPS, alpha_grow, stage_schedule
  sch = stage_schedule(4, 32, {4: 16, 8: 16, 16: 8, 32: 8}, 300000)
  assert [s[0] for s in sch] == ['4', '4to8', '8', '8to16', '16', '16to32', '32']
  assert [s[2] for s in sch] == [False, True, False, True, False, True, False]
  assert sch[0][4] == 18750 and sch[3][4] == 37500 and sch[-1][4] == LAST_STAGE_STEPS and sch[-2][4] == 37500
  full = stage_schedule()
  assert len(full) == 13 and full[-1][:3] == ('256', 256, False) and full[-2][0] == '128to256'
  assert alpha_grow(0, 100) == 0.0 and alpha_grow(50, 100) == 0.5


def test_stage_schedule_matches_what_pggan_runner_sets():
  """tests/golden/stage_driver.json = the per-stage flags pggan_runner.main itself set when it was executed by
  oracle/ref_runner.run_stage_driver (tools/make_golden.py); where /root/reference is mounted it is run again."""
  import json
  import os
  from twingan_amd.runner import stage_schedule
  with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'stage_driver.json')) as fh:
    cases = json.load(fh)
  assert len(cases) >= 3
  for c in cases:
    table = {int(k): v for k, v in c['hw_to_batch_size'].items()}
    mine = stage_schedule(c['start_hw'], c['max_hw'], table, c['num_images_per_resolution'])
    ref = c['stages']
    assert [(s['name'], s['hw'], s['is_growing'], s['batch_size'], s['max_number_of_steps']) for s in ref] == mine
    # warm start: every stage restores from the previous one, ignoring missing variables exactly when growing
    assert [s['checkpoint_path'] for s in ref] == [None] + [s['name'] for s in ref[:-1]]
    assert all(s['ignore_missing_vars'] == s['is_growing'] for s in ref)
    if os.path.isdir('/root/reference'):
      from oracle import ref_runner
      assert ref_runner.run_stage_driver(c['start_hw'], c['max_hw'], table, c['num_images_per_resolution']) == ref


def test_warm_start_ignores_missing_and_reshaped_vars():
  """ignore_missing_vars semantics (pggan_runner.py:136-146): shared blocks are copied, the new resolution's layers
  (and the from_rgb / to_rgb of the new size) keep their fresh initialisation."""
  from twingan_amd import Config
  from twingan_amd.params import ParamStore, declare_twingan
  from twingan_amd.runner import warm_start
  prev = declare_twingan(ParamStore('cpu'), Config(hw=8, max_ch=8)).build(seed=1)
  class T:      # the part of Trainer that warm_start touches
    store = declare_twingan(ParamStore('cpu'), Config(hw=16, max_ch=8, is_growing=True)).build(seed=2)
  loaded = warm_start(T, prev.state_dict())
  assert 'generator/block_8x8x8/Conv/weights' in loaded and 'encoder_content/from_rgb_8x8/Conv/weights' in loaded
  assert 'generator/block_16x16x8/Conv/weights' not in loaded
  a, b = prev.state_dict(), T.store.state_dict()
  assert torch.equal(a['generator/block_8x8x8/Conv/weights'], b['generator/block_8x8x8/Conv/weights'])
  assert torch.equal(a['generator/generator_to_rgb_8x8/Conv/weights'], b['generator/generator_to_rgb_8x8/Conv/weights'])


def test_state_dict_carries_non_trainable_variables():
  """Moving / renorm statistics and spectral-norm vectors travel with state_dict(include_state=True) and back."""
  from twingan_amd import Config
  from twingan_amd.params import ParamStore, declare_twingan
  cfg = Config(hw=8, max_ch=8, generator_norm_type='batch_renorm', spectral_norm=True)
  a = declare_twingan(ParamStore('cpu'), cfg).build(seed=1)
  b = declare_twingan(ParamStore('cpu'), cfg).build(seed=2)
  k_bn = 'generator/block_4x4x8/Conv/BatchNorm/renorm_mean_s'
  k_u = 'discriminator_s/from_rgb_8x8/Conv/u'
  a.state[k_bn].fill_(0.25)
  sd = a.state_dict(include_state=True)
  assert k_bn in sd and k_u in sd and 'renorm/rmax' not in sd
  assert k_bn not in a.state_dict()
  b.load_state_dict(sd)
  assert torch.equal(b.state[k_bn], a.state[k_bn]) and torch.equal(b.state[k_u], a.state[k_u])


def test_bench_line_contract():
  """The JSON line bench.py printed for the final build of the round (profiles/r01_j_bench.json) carries every field of
  the driver's contract, the metric of BASELINE.json, and a self-consistent roofline / cpu_baseline; the command line
  takes --gpus / --steps / --warmup."""
  import json
  import os
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  with open(os.path.join(root, 'profiles', 'r01_j_bench.json')) as fh:
    d = json.load(fh)
  with open(os.path.join(root, 'BASELINE.json')) as fh:
    base = json.load(fh)
  for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
            'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
    assert k in d, k
  sys.path.insert(0, root)
  import bench
  assert bench.METRIC == base['metric']      # the line's metric is BASELINE.json's, verbatim
  assert d['unit'] == 'images/sec' and d['higher_is_better'] is True
  assert d['n_gpus'] == 1 and d['scaling'] == 'weak' and d['dtype'] == 'bf16' and d['data'] == 'synthetic'
  assert d['vs_baseline'] is None and not base['published']      # no published number for this metric
  assert 'workload' in d['config'] and 'model' not in d['config']
  assert abs(d['value'] - d['config']['global_batch'] / (d['ms_per_step'] * 1e-3)) < 0.01 * d['value']
  r = d['roofline']
  assert r['bound'] in ('hbm', 'mfma') and r['unit'] in ('GB/s', 'TFLOP/s') and r['peak'] in (8000.0, 2500.0)
  assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3 and r['traffic'] is None or r['traffic'] > 0
  c = d['cpu_baseline']
  assert c['kind'] in ('port', 'reference') and c['cores'] >= 1 and c['value'] > 0 and c['unit'] == d['unit'] and c['sample']
  src = open(os.path.join(root, 'bench.py')).read()
  for flag in ('--gpus', '--steps', '--warmup'):
    assert "'%s'" % flag in src


def test_bench_line_stays_small_and_names_the_dominant_family():
  """Round 5's line was 20.6 KB and the driver recorded `parsed: null`.  The launch records of a roofline pass are rebuilt
  from the committed per-shape table of that round (profiles/r05_z_shapes_eager_step.json: launches, mean duration, rates
  per (entry point, layer shape, n)) with the kernel symbols the dispatch test pins (tests/golden/bench_dispatch_kernels.json);
  bench.summarize_launches must give a compact object -- scalars + six family rows, every template instantiation of a kernel in
  ONE family -- and the whole line must stay below bench.MAX_LINE_BYTES with the tables in a side file."""
  import json
  import re
  import sys
  import tempfile
  sys.path.insert(0, ROOT)
  import bench
  rows = json.load(open(os.path.join(ROOT, 'profiles', 'r05_z_shapes_eager_step.json')))
  pins = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'bench_dispatch_kernels.json')))
  rec = []
  for r in rows:
    m = re.match(r'(\w+)\[(.*)\]$', r['kernel'])
    name, tag = (m.group(1), m.group(2)) if m else (r['kernel'], '')
    ms = r['avg_us'] * 1e-3
    kname = ''
    if name.startswith('tg_conv2d'):
      parts = tag.split(':')
      shape = ':'.join(p for p in parts if re.match(r'(k\d|c\d|hw\d)', p))
      n = next((p[1:] for p in parts if re.match(r'n[\d+]+$', p)), '')
      kname = pins.get(shape, {}).get(name, {}).get(n, 'conv_tile_kernel<3,32,32,1>')
    for _ in range(2 * r['launches']):      # two instrumented steps
      rec.append((name, tag, r['tflops'] * 1e12 * ms * 1e-3, r['gbs'] * 1e9 * ms * 1e-3, ms, kname))
  pmc = bench.load_pmc()
  roof, tables = bench.summarize_launches(rec, 2, pmc=pmc)
  assert roof['bound'] in ('hbm', 'mfma') and abs(roof['frac'] - roof['achieved'] / roof['peak']) < 1e-3
  assert len(roof['families']) <= 6 and all('<' not in f['kernel'] for f in roof['families'])
  assert roof['kernel'] == roof['families'][0]['kernel']
  # the forward / backward-data tile kernel is what the step spends most time in once its instantiations are one family
  assert roof['kernel'] == 'conv_tile_kernel', roof['kernel']
  assert 0.0 < roof['conv_mfma_util_time_weighted'] < 1.0 and 0.0 < roof['step_roofline_frac'] < 1.0
  assert not any(isinstance(v, (list, dict)) for k, v in roof.items() if k != 'families')
  assert len(rows) - 4 <= len(tables['shapes']) <= len(rows) and len(tables['symbols']) >= len(tables['families'])
  line = dict(metric=bench.METRIC, value=1032.2, unit='images/sec', n_gpus=1, steps=20, warmup=5, ms_per_step=15.5,
              higher_is_better=True, scaling='weak', vs_baseline=None, dtype='bf16', data='synthetic',
              config={'workload': 'x' * 300, 'global_batch': 16}, roofline=roof,
              cpu_baseline=dict(value=0.9, unit='images/sec', cores=32, kind='port', sample='y' * 200,
                                literal_schedule=dict(value=0.4, unit='images/sec', sample='z' * 150)),
              timed_regions=dict(repeats=3, steps_each=20, reported='median', value_max=1.0, value_min=1.0))
  assert len(json.dumps(line)) < bench.MAX_LINE_BYTES < 8192, len(json.dumps(line))
  with tempfile.TemporaryDirectory() as td:
    os.environ['TG_BENCH_TABLES'] = os.path.join(td, 't.json')
    try:
      path = bench.write_tables(tables, 3)
    finally:
      del os.environ['TG_BENCH_TABLES']
    assert json.load(open(path))['shapes'][0]['kernel'] == tables['shapes'][0]['kernel']


def test_bench_gpus_n_spawns_n_ranks():
  """`python bench.py --gpus 2` with no launcher in the environment starts 2 ranks itself (the reference's
  one-process --num_clones=N, deployment/model_deploy.py:186-239, as one process per GPU).  --launch-check runs the
  launcher, the process group (gloo here, RCCL on a GPU box) and the per-segment all-reduce schedule without kernels."""
  import json
  import subprocess
  import sys
  env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
  env['CUDA_VISIBLE_DEVICES'] = env['HIP_VISIBLE_DEVICES'] = ''      # a GPU box has one GPU: force the gloo path
  out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--hw', '32',
                        '--max-ch', '16', '--launch-check'], capture_output=True, text=True, timeout=600, env=env)
  assert out.returncode == 0, out.stderr[-2000:]
  lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
  assert len(lines) == 1, out.stdout      # rank 0 only
  d = json.loads(lines[0])
  assert d['n_gpus'] == 2 and d['config']['parallelism'] == 'dp2' and d['config']['global_batch'] == 32
  assert 'world 2' in d['config']['collective'] and d['scaling'] == 'weak'
  lc = d['launch_check']
  seg = lc['segments']
  assert len(seg['g']) >= 1 and len(seg['d']) >= 1
  # the diagnostics the first real N > 1 run will be read by: who took part, the bytes through the collective per step
  # (= both groups' flat gradient buffers, once each), one collective per backward segment, the exposed wait
  assert lc['rccl_world'] == 2
  assert lc['allreduce_bytes_per_step'] == lc['grad_bytes']['g'] + lc['grad_bytes']['d']
  assert lc['collectives_per_step'] == len(seg['g']) + len(seg['d'])
  assert lc['exposed_allreduce_ms'] >= 0.0


def test_grad_reducer_statistics_single_process():
  """dp.GradReducer.stats(): bytes / collectives / finishes / exposed wait of the per-segment all-reduces, exercised with
  a one-rank gloo group and always=True (the tools/rccl_smoke.py path: a one-rank sum is the identity)."""
  import torch.distributed as dist
  from twingan_amd.dp import GradReducer
  import socket
  sk = socket.socket(); sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]; sk.close()
  dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1)
  try:
    red = GradReducer(1, None, always=True)
    assert red.active
    buf = torch.arange(1000, dtype=torch.float32)
    for _ in range(3):
      red.start(buf[:600], n_buckets=1)
      red.start(buf[600:], n_buckets=2)
      red.finish()
    st = red.stats()
    assert st['allreduce_bytes'] == 3 * 4000 and st['collectives'] == 3 * 3 and st['finishes'] == 3
    assert st['exposed_allreduce_ms'] >= 0.0 and st['exposed_max_ms'] >= st['exposed_allreduce_ms']
    assert torch.equal(buf, torch.arange(1000, dtype=torch.float32))
    red.reset_stats()
    assert red.stats()['allreduce_bytes'] == 0
    assert GradReducer(1, None).start(buf) == 0      # a single clone without `always`: no collective, no statistics
  finally:
    dist.destroy_process_group()


def test_gradient_phases_are_contiguous_ranges():
  """params.grad_phase: the flat gradient buffer is laid out segment by segment (what one backward segment completes is
  one contiguous range = one all-reduce), the initial values do not depend on the layout; growing stages are laid out the
  same way with the shrink path in the high phase."""
  from twingan_amd import Config
  from twingan_amd.params import ParamStore, declare_twingan, grad_phase
  cfg = Config(hw=128, max_ch=64)
  st = declare_twingan(ParamStore('cpu'), cfg).build(seed=3)
  assert sorted(st.phase_bounds['g']) == [0, 1, 2] and sorted(st.phase_bounds['d']) == [0, 1]
  for g in ('g', 'd'):
    b = [st.phase_bounds[g][p] for p in sorted(st.phase_bounds[g])]
    assert b[0][0] == 0 and b[-1][1] == st.grad[g].numel()
    assert all(x[1] == y[0] and x[0] % 64 == 0 for x, y in zip(b, b[1:]))
    for k, s in st.specs.items():
      if s['group'] == g:
        lo, hi = st.phase_bounds[g][st.phase[k]]
        assert lo <= st.offsets[k] < hi, k
  assert grad_phase('generator/block_128x128x32/Conv/weights', cfg) == 0
  assert grad_phase('encoder_content/encoder_block_32x32x64/Conv_1/InstanceNorm/gamma_s', cfg) == 1
  assert grad_phase('encoder_content/encoder_block_64x64x64/Conv/weights', cfg) == 2
  assert grad_phase('encoder_content/from_rgb_128x128/Conv/weights', cfg) == 2
  assert grad_phase('discriminator_t/encoder_block_64x64x64/Conv/biases', cfg) == 1
  assert grad_phase('discriminator_t/before_fc_1x1x64/Conv_1/weights', cfg) == 0
  assert grad_phase('discriminator_s/prediction/fully_connected/weights', cfg) == 0
  # same seed, layout switched off: identical values under every name
  flat = declare_twingan(ParamStore('cpu'), cfg)
  flat.phase_of = None
  flat = flat.build(seed=3)
  a, b = st.state_dict(), flat.state_dict()
  assert all(torch.equal(a[k], b[k]) for k in a)
  assert flat.offsets != st.offsets
  # growing stages are split too: the shrink path's fromRGB (hw / 2) is blended in above the cut, so it completes with the
  # full-resolution block whatever its own resolution says
  gcfg = Config(hw=64, max_ch=16, is_growing=True)      # overlap_cut_hw 32 = hw / 2
  grow = declare_twingan(ParamStore('cpu'), gcfg).build(seed=0)
  assert sorted(grow.phase_bounds['g']) == [0, 1, 2] and sorted(grow.phase_bounds['d']) == [0, 1]
  assert grad_phase('encoder_content/from_rgb_32x32/Conv/weights', gcfg) == 2
  assert grad_phase('discriminator_s/from_rgb_32x32/Conv/biases', gcfg) == 1
  assert grad_phase('encoder_content/encoder_block_32x32x16/Conv/weights', gcfg) == 1
  assert grad_phase('generator/generator_to_rgb_32x32/Conv/weights', gcfg) == 0
  one = declare_twingan(ParamStore('cpu'), Config(hw=32, max_ch=16, is_growing=True)).build(seed=0)      # hw <= cut: one phase
  assert list(one.phase_bounds['g']) == [0] and list(one.phase_bounds['d']) == [0]


def test_segmented_backward_equals_plain_backward():
  """ops.Cuts: stopping the backward at detached leaves and resuming from their producers, segment by segment, leaves
  the same parameter gradients as one loss.backward() (plain torch graph: the mechanism has no kernels of its own)."""
  from twingan_amd.ops import Cuts
  torch.manual_seed(0)
  w = [torch.randn(6, 6, dtype=torch.float64, requires_grad=True) for _ in range(4)]
  x = torch.randn(5, 6, dtype=torch.float64)

  def net(cut):
    h1 = torch.tanh(x @ w[0])                   # "high-resolution encoder"
    h1c = Cuts.cut(h1, 2) if cut else h1        # inner cut: resumes in segment 2
    h2 = torch.tanh(h1c @ w[1])                 # "low-resolution encoder"
    skip = Cuts.cut(h1, 2) if cut else h1       # skip edge out of the high part
    e = Cuts.cut(h2, 1) if cut else h2          # content edge
    g = torch.tanh(e @ w[2]) + skip @ w[3]      # "generator"
    return (g ** 2).sum() + 0.1 * (h2.detach() * e).sum()

  net(False).backward()
  want = [p.grad.clone() for p in w]
  for p in w:
    p.grad = None
  Cuts.begin()
  try:
    net(True).backward()
    assert w[2].grad is not None and w[0].grad is None and w[1].grad is None      # segment 0 stops at the cuts
    for seg in (1, 2):
      roots, grads = Cuts.roots(seg)
      assert roots
      torch.autograd.backward(roots, grads)
      assert (w[1].grad is not None) and ((w[0].grad is not None) == (seg == 2))
  finally:
    Cuts.end()
  for p, g in zip(w, want):
    assert torch.allclose(p.grad, g, rtol=1e-12, atol=1e-14)
  y = torch.ones(2, requires_grad=True)
  assert Cuts.cut(y, 1) is y                    # inactive: the identity


def test_tf1_bilinear_resize_restated():
  """inference.resize_bilinear_tf1 = tf.image.resize_images(BILINEAR, align_corners=False) of TF 1.8
  (inference/image_translation_infer.py:58): source coordinate = index * in / out, neighbours clamped."""
  import numpy as np
  from twingan_amd.inference import resize_bilinear_tf1

  def ref(img, hw):
    b, h, w, c = img.shape
    out = np.zeros((b, hw, hw, c))
    for y in range(hw):
      sy = y * h / hw
      y0 = int(np.floor(sy)); y1 = min(y0 + 1, h - 1); fy = sy - y0
      for x in range(hw):
        sx = x * w / hw
        x0 = int(np.floor(sx)); x1 = min(x0 + 1, w - 1); fx = sx - x0
        top = img[:, y0, x0] * (1 - fx) + img[:, y0, x1] * fx
        bot = img[:, y1, x0] * (1 - fx) + img[:, y1, x1] * fx
        out[:, y, x] = top * (1 - fy) + bot * fy
    return out
  rng = np.random.RandomState(0)
  for h, w, hw in ((5, 7, 8), (16, 12, 8), (8, 8, 8), (3, 3, 16)):
    a = rng.rand(2, h, w, 3).astype(np.float32)
    assert np.abs(resize_bilinear_tf1(torch.from_numpy(a), hw).numpy() - ref(a, hw)).max() < 1e-6


def test_network_registry_has_the_reference_entries():
  """twingan.GanModel._select_network (twingan.py:122-140) returns seven functions; the product binds all of them."""
  from twingan_amd import pggan
  from twingan_amd.twingan import select_network
  reg = select_network('pggan')
  assert sorted(reg) == sorted(['generator_network_fn', 'discriminator_network_fn', 'encoder_network_fn',
                                'encoder_style_network_fn', 'encoder_classification_fn', 'encoder_distillation_fn',
                                'get_noise_shape'])
  assert reg['encoder_distillation_fn'] is reg['encoder_classification_fn'] is pggan.encoder_classification
  assert reg['get_noise_shape'](16, 256) == (16, 1, 1, 256) and reg['get_noise_shape'](None, 16) == (None, 1, 1, 16)
  with pytest.raises(NotImplementedError):
    select_network('cyclegan')


def test_flash_attention_host_queries():
  """The host-only entry points of the flash-attention family answer without a GPU: the supported-shape rule of
  include/twingan_hip.h and workspace sizes that grow with the pass (forward < backward < backward of the backward), are
  256-byte multiples and scale with the batch."""
  from twingan_amd import _lib
  lib = _lib.load()
  ok = lambda ln, dk, dv: bool(lib.tg_flash_attention_supported(ln, dk, dv))
  assert ok(4096, 8, 64) and ok(256, 16, 128) and ok(128, 8, 256)
  assert not ok(64, 8, 64) and not ok(4096, 4, 64) and not ok(4096, 8, 32) and not ok(4000, 8, 64)
  ws = lambda n, which, dk=8, dv=64: int(lib.tg_flash_attention_workspace_bytes(n, 4096, dk, dv, which))
  f, b, bb = ws(16, 0), ws(16, 1), ws(16, 2)
  assert 0 < f < b < bb and all(x % 256 == 0 for x in (f, b, bb))
  assert f >= 16 * 4096 * 64 * 2 and ws(32, 2) == 2 * bb          # one packed copy of v at least; linear in n
  assert ws(16, 1, dk=16) < b + 1 and ws(16, 0, dk=4) == 0         # d_qk = 16 needs no padded q / k copies; unsupported: 0


def test_native_normalisers_refuse_what_the_reference_refuses():
  """nets/pggan_utils.py:177,191: tf.contrib's layers take no conditional layer ('Tensorflow implementation does not support
  `conditional_layer`'); and the plain PGGAN trainer's empty postfix would make the layer's scope the empty string --
  refused rather than guessed."""
  from twingan_amd import Config
  from twingan_amd.params import ParamStore, declare_pggan, declare_twingan, norm_var, NATIVE_NORM
  for norm in ('batch_renorm_native', 'layer_norm_native'):
    with pytest.raises(NotImplementedError):
      declare_twingan(ParamStore('cpu'), Config(hw=8, max_ch=8, generator_norm_type=norm, use_style_embedding=True,
                                                style_embed_size=4))
    with pytest.raises(NotImplementedError):
      declare_pggan(ParamStore('cpu'), Config(hw=8, max_ch=8, generator_norm_type=norm))
    store = declare_twingan(ParamStore('cpu'), Config(hw=8, max_ch=8, generator_norm_type=norm)).build(seed=0)
    assert 'generator/block_4x4x8/Conv/_s/gamma' in store.specs and 'generator/block_4x4x8/Conv/_t/beta' in store.specs
    has_state = any('/_s/renorm_mean_weight' in k for k in store.state)
    assert has_state == (norm == 'batch_renorm_native')
  assert norm_var('a/Conv', NATIVE_NORM, 'gamma', 's') == 'a/Conv/_s/gamma'
  assert norm_var('a/Conv', 'BatchNorm', 'gamma', 's') == 'a/Conv/BatchNorm/gamma_s'
  assert norm_var('a/Conv', 'InstanceNorm', 'beta', '') == 'a/Conv/InstanceNorm/beta'
  with pytest.raises(NotImplementedError):
    declare_twingan(ParamStore('cpu'), Config(hw=8, max_ch=8, generator_norm_type='group_norm'))
