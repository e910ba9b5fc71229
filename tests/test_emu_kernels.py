"""CPU: the HIP kernels' own source, compiled for the host over tests/hipemu (fibers for threads, the gfx950 lane layouts of
MFMA / DPP / permlane swaps / transposing LDS reads emulated), run through the product's Python path against the same
oracles as on the GPU.  What this checks without a GPU is the kernels' LOGIC -- index arithmetic, LDS staging, lane exchanges,
operand layouts, epilogues, the host-side dispatch -- not timing and not the hardware's rounding inside an MFMA.

The subset in tests/hipemu/fast_subset.txt is every case of tests/test_gpu_ops.py that runs in under a second over the
emulated kernels (all operator families; the large shapes are left to `TG_EMU=1 python -m pytest tests/test_gpu_ops.py -m gpu`,
which passes all but four of its cases -- those compare device libm bit patterns, the CUDA-tensor check and RCCL).  It
runs in a child process: the harness redirects torch's CUDA entry points process-wide.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.join(ROOT, 'tests', 'hipemu')


def _run(args, timeout, **extra):
  env = dict(os.environ, TG_EMU='1', **extra)
  env.pop('TG_LIB_PATH', None)
  return subprocess.run([sys.executable, '-m', 'pytest', '-m', 'gpu', '-q', '-x', '-p', 'no:cacheprovider', '--tb=short'] + args,
                        cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)


def test_emulated_library_builds_and_exports_the_c_abi():
  """The emulated library is the product's source behind the product's ABI: every entry point include/twingan_hip.h
  declares (and twingan_amd._lib binds) is there."""
  import ctypes
  sys.path.insert(0, HERE)
  try:
    import build
  finally:
    sys.path.pop(0)
  lib = ctypes.CDLL(build.build())
  from twingan_amd import _lib
  missing = [name for name in _lib.SIGNATURES if not hasattr(lib, name)]
  assert not missing, missing


def test_operator_parity_tests_pass_over_the_emulated_kernels():
  ids = [ln.strip() for ln in open(os.path.join(HERE, 'fast_subset.txt')) if ln.strip()]
  assert len(ids) > 100
  r = _run(ids, timeout=1500)
  tail = r.stdout[-3000:]
  assert r.returncode == 0, tail
  assert ' passed' in tail and 'failed' not in tail and 'skipped' not in tail.split('passed')[-1], tail


def test_unpooling_backward_data_over_the_emulated_kernels():
  """This round's kernel path, developed against the emulation first: the cases small enough for it."""
  r = _run(['tests/test_gpu_ops.py', '-k', 'unpool and (0-dtype or 1-dtype or 7-dtype or block_end)'], timeout=1500)
  assert r.returncode == 0, r.stdout[-3000:]


def test_thin_output_kernels_over_the_emulated_kernels():
  """conv_thin16_kernel at the smallest shapes that dispatch it, and the multi-kernel spectral-norm launch."""
  r = _run(['tests/test_gpu_ops.py', '-n', '3', '-k',
            '(thin_output_kernel_matches and 16-16-dtype0) or spectral_norm_of_many'], timeout=2400)
  assert r.returncode == 0, r.stdout[-3000:]
  assert ' passed' in r.stdout[-3000:]
