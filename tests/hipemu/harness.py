"""Runs the product's Python path over the EMULATED kernels (tests/hipemu/libtwingan_emu.so) on CPU tensors.

TEST INFRASTRUCTURE ONLY.  ``enable()``:
  * builds the emulated library if needed and points ``twingan_amd._lib`` at it (same C ABI, same ctypes table);
  * replaces the two device checks of ``twingan_amd.ops`` (``_stream`` -> the null stream, ``_chk`` -> contiguity only);
  * installs a TorchFunctionMode that sends every ``device='cuda...'`` of a torch call to the CPU and turns the
    ``torch.cuda`` stream / synchronisation calls the tests make into no-ops,
so that the GPU parity tests (tests/test_gpu_*.py) can be executed, unmodified, against the same HIP source compiled for the
host: ``TG_EMU=1 python -m pytest tests/test_gpu_ops.py -m gpu -k ...``.  What that checks is the kernels' LOGIC (indexing, LDS
staging, lane exchanges, MFMA operand layouts, epilogues) -- not timing, not the hardware's rounding of MFMA sums.
"""
import contextlib
import os
import sys

import torch
from torch.overrides import TorchFunctionMode

HERE = os.path.dirname(os.path.abspath(__file__))
_enabled = False


def _is_cuda_dev(v):
  if isinstance(v, torch.device):
    return v.type == 'cuda'
  return isinstance(v, str) and v.startswith('cuda')


class _CpuForCuda(TorchFunctionMode):
  def __torch_function__(self, func, types, args=(), kwargs=None):
    kwargs = dict(kwargs or {})
    name = getattr(func, '__name__', '')
    if name == 'cuda' and args and isinstance(args[0], torch.Tensor):      # Tensor.cuda()
      return args[0]
    if 'device' in kwargs and _is_cuda_dev(kwargs['device']):
      kwargs['device'] = torch.device('cpu')
    if any(_is_cuda_dev(a) for a in args):
      args = tuple(torch.device('cpu') if _is_cuda_dev(a) else a for a in args)
    return func(*args, **kwargs)


class _NullStream:
  cuda_stream = 0

  def wait_stream(self, other):
    pass

  def wait_event(self, ev):
    pass

  def synchronize(self):
    pass

  def record_event(self, ev=None):
    return ev


class _NullEvent:
  def __init__(self, *a, **k):
    pass

  def record(self, *a):
    pass

  def synchronize(self):
    pass

  def elapsed_time(self, other):
    return 0.0

  def wait(self, *a):
    pass


def library_path():
  sys.path.insert(0, HERE)
  try:
    import build as _build
  finally:
    sys.path.pop(0)
  return _build.build()


def enable():
  """Idempotent.  Returns the path of the emulated library."""
  global _enabled
  path = library_path()
  if _enabled:
    return path
  from twingan_amd import _lib
  _lib.LIB_PATH = path
  _lib._lib = None
  from twingan_amd import ops

  def _chk(*ts):
    for t in ts:
      if t is not None and not t.is_contiguous():
        raise _lib.TgError('twingan_amd ops need contiguous NHWC tensors')
  ops._stream = lambda: 0
  ops._chk = _chk
  null = _NullStream()
  torch.cuda.current_stream = lambda device=None: null
  torch.cuda.Stream = lambda *a, **k: _NullStream()
  torch.cuda.Event = _NullEvent
  torch.cuda.stream = lambda s: contextlib.nullcontext()
  torch.cuda.synchronize = lambda device=None: None
  torch.cuda.device = lambda d: contextlib.nullcontext()
  torch.cuda.current_device = lambda: 0
  torch.Tensor.record_stream = lambda self, stream: None
  torch.cuda.manual_seed = lambda s: torch.manual_seed(s)
  torch.cuda.get_rng_state = lambda device=None: torch.get_rng_state()
  torch.cuda.set_rng_state = lambda st, device=None: torch.set_rng_state(st)
  _CpuForCuda().__enter__()      # for the life of the process
  _enabled = True
  return path
