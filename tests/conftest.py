import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


# TG_EMU=1: run the GPU parity tests on CPU over the emulated kernels (tests/hipemu: the HIP sources compiled for the host) --
# a development check of the kernels' logic when no GPU is at hand; never set on the GPU box
EMU = os.environ.get('TG_EMU') == '1'


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
  if EMU:
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'hipemu'))
    import harness
    harness.enable()


# GPU run order: primitives first, then the kernels at the bench's shapes, the data kernels, the golden fixtures, and the
# whole-model tests last -- under `-x` a failing end-to-end test must not hide the 180+ kernel parity tests behind it
# (round 2: one model test stopped the driver's run before tests/test_gpu_ops.py had started)
_ORDER = ('test_gpu_ops.py', 'test_gpu_bench_shapes.py', 'test_gpu_data.py', 'test_golden.py', 'test_gpu_model.py')


def pytest_collection_modifyitems(config, items):
  def rank(item):
    name = os.path.basename(str(item.fspath))
    return _ORDER.index(name) if name in _ORDER else -1      # CPU-side files keep their place in front
  items.sort(key=rank)                                        # stable: the order inside a file is untouched
  import torch
  if EMU:      # what the host cannot stand in for: device libm bit patterns, the CUDA-tensor check, RCCL
    no = pytest.mark.skip(reason='not meaningful over the emulated kernels')
    for item in items:
      if item.name.split('[')[0] in ('test_adam_kernel_tf_semantics', 'test_adam_device_tick_matches_host_schedule',
                                     'test_errors_are_loud', 'test_rccl_allreduce_wrapper_single_rank') or \
          os.path.basename(str(item.fspath)) == 'test_gpu_data.py' or 'graph_replay' in item.name or \
          item.name.split('[')[0].endswith('_and_graph') or \
          item.name.startswith(('test_data_parallel_two_clones', 'test_deterministic_mode_makes_16_bit', 'test_rccl_segmented_capture')):
        # the input pipeline asks torch for a GPU itself; hipGraph capture (also inside the determinism test); RCCL
        item.add_marker(no)
    return
  if torch.cuda.is_available():
    return
  skip = pytest.mark.skip(reason='no GPU visible')
  for item in items:
    if 'gpu' in item.keywords:
      item.add_marker(skip)
