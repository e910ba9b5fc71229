"""Builds tests/hipemu/libtwingan_emu.so: twingan_amd/csrc/*.hip compiled as host C++ over the hipemu stand-in for the HIP
runtime (hip/hip_runtime.h, emu.cpp), same C ABI as libtwingan_hip.so.  TEST INFRASTRUCTURE: lets CPU tests run the kernels'
own source (index arithmetic, LDS staging, lane exchanges, MFMA layouts) against the oracle; never loaded by the product.

  python tests/hipemu/build.py [--force]     -> path of the library (rebuilt when a source is newer)
"""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'twingan_amd', 'csrc')
OUT = os.path.join(HERE, '_build')
LIB = os.path.join(HERE, 'libtwingan_emu.so')
CXX = os.environ.get('HIPEMU_CXX', '/opt/rocm/lib/llvm/bin/clang++')
# comm.hip (RCCL) is replaced by stubs in emu.cpp
SOURCES = ['capi', 'conv_direct', 'conv_mfma', 'conv_tile', 'conv_wgrad_tile', 'conv_small', 'conv_img', 'pointwise', 'norm', 'reduce',
           'sn', 'attention', 'preprocess', 'flash']
# -fmax-type-align=4: a 16-byte vector access at a 4-byte-aligned address is legal on the GPU (x86 would fault on movaps);
# -ffp-contract=fast (+ -mfma where the host has it): hipcc contracts a * b + c into fma by default, and the fp32 parity
# path's model-level tests sit on LeakyReLU units that one last-bit difference flips
FLAGS = ['-std=c++17', '-O1', '-fPIC', '-fno-strict-aliasing', '-fmax-type-align=4', '-ffp-contract=fast', '-w', '-I', HERE, '-I', CSRC,
         '-DHIPEMU=1']
try:
  if ' fma ' in open('/proc/cpuinfo').read():
    FLAGS.insert(6, '-mfma')
except OSError:
  pass

_DYN = re.compile(r'extern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?([A-Za-z_][A-Za-z0-9_ ]*?)\s+([A-Za-z_][A-Za-z0-9_]*)\[\];')


def translate(name):
  """The one construct a macro cannot reach: `extern __shared__ T name[];` becomes a pointer to the emulator's dynamic LDS."""
  src = open(os.path.join(CSRC, name + '.hip')).read()
  src = _DYN.sub(lambda m: '%s* const %s = (%s*)hipemu::dyn_lds;' % (m.group(1), m.group(2), m.group(1)), src)
  dst = os.path.join(OUT, name + '.cpp')
  if not os.path.exists(dst) or open(dst).read() != src:
    open(dst, 'w').write(src)
  return dst


def stale(target, deps):
  if not os.path.exists(target):
    return True
  t = os.path.getmtime(target)
  return any(os.path.getmtime(d) > t for d in deps)


def compile_one(job):
  src, obj, deps, force = job
  if force or stale(obj, deps):
    subprocess.check_call([CXX] + FLAGS + ['-c', src, '-o', obj])
    return True
  return False


def build(force=False):
  os.makedirs(OUT, exist_ok=True)
  common = [os.path.join(CSRC, 'tg_common.h'), os.path.join(ROOT, 'include', 'twingan_hip.h'), os.path.join(HERE, 'hip', 'hip_runtime.h'),
            os.path.abspath(__file__)]
  jobs = []
  for name in SOURCES:
    jobs.append((translate(name), os.path.join(OUT, name + '.o'), [os.path.join(CSRC, name + '.hip')] + common, force))
  jobs.append((os.path.join(HERE, 'emu.cpp'), os.path.join(OUT, 'emu.o'), [os.path.join(HERE, 'emu.cpp')] + common, force))
  with ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as ex:
    changed = list(ex.map(compile_one, jobs))
  if any(changed) or not os.path.exists(LIB):
    tmp = LIB + '.tmp%d' % os.getpid()      # linked aside, then renamed: processes that have the old file mapped keep it
    subprocess.check_call([CXX, '-shared', '-fPIC'] + [j[1] for j in jobs] + ['-o', tmp])
    os.replace(tmp, LIB)
  return LIB


if __name__ == '__main__':
  print(build(force='--force' in sys.argv))
