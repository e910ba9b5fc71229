"""TEST INFRASTRUCTURE -- imports modules of the reference tree (read-only, /root/reference) into this Python-3
process without writing anything there and without copying them here:

  * the source is read from the reference tree at import time, passed through lib2to3 in memory (print statements,
    dict.iteritems, implicit relative imports ...), and -- for files WITHOUT `from __future__ import division` --
    every `/` is rewritten to Python-2 semantics (floor division between integers), because e.g. the channel
    schedule of nets/pggan_utils.py:369-372 relies on it;
  * modules the hot path never executes (datasets, preprocessing, deployment, the slim classifier zoo, image IO)
    resolve to absorbing stubs.
"""
import ast
import importlib.abc
import importlib.machinery
import os
import sys
import warnings

import numpy as np

from . import tfapi
from .core import Dimension

REAL = ('nets', 'nets.pggan', 'nets.pggan_utils', 'libs', 'libs.ops', 'libs.batch_norm', 'libs.instance_norm',
        'libs.sn', 'libs.self_attention', 'libs.gdrop', 'util_misc', 'twingan', 'image_generation', 'model',
        'model.model_inheritor', 'pggan_runner', 'deployment', 'deployment.model_deploy')
STUBS = ('datasets', 'preprocessing', 'util_io', 'nets.cyclegan', 'nets.cyclegan_dis',
         'nets.nets_factory', 'scipy.misc')      # (PIL is installed here: util_misc.py imports the real one)
# real modules inside a stubbed package (the TwinGAN trainer's own preprocessing; the factory and the classifier
# zoo's preprocessing stay stubs).  Import preprocessing_util before danbooru_preprocessing: the latter fetches it as an
# attribute of the (stub) package.
REAL_IN_STUBS = ('preprocessing.preprocessing_util', 'preprocessing.danbooru_preprocessing')


def py2div(a, b):
  ints = (int, np.integer, Dimension)
  if isinstance(a, ints) and isinstance(b, ints) and not isinstance(a, bool) and not isinstance(b, bool):
    if isinstance(a, Dimension) or isinstance(b, Dimension):
      return Dimension(int(a) // int(b))
    return a // b
  return a / b


class _Py2Div(ast.NodeTransformer):
  def visit_BinOp(self, node):
    self.generic_visit(node)
    if isinstance(node.op, ast.Div):
      return ast.copy_location(ast.Call(ast.Name('__py2div__', ast.Load()), [node.left, node.right], []), node)
    return node

  def visit_AugAssign(self, node):
    self.generic_visit(node)
    if isinstance(node.op, ast.Div):
      load = ast.parse(ast.unparse(node.target), mode='eval').body
      call = ast.Call(ast.Name('__py2div__', ast.Load()), [load, node.value], [])
      return ast.copy_location(ast.Assign([node.target], call), node)
    return node


def _to_py3(src, path, top_level):
  with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    from lib2to3 import refactor
    fixers = refactor.get_fixers_from_package('lib2to3.fixes')
    if top_level:      # a sibling `import x` of a top-level script is already absolute
      fixers = [f for f in fixers if f != 'lib2to3.fixes.fix_import']
    tool = refactor.RefactoringTool(fixers)
    return str(tool.refactor_string(src if src.endswith('\n') else src + '\n', path))


class _ReferenceFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
  def __init__(self, root):
    self.root = root

  def _path(self, name):
    base = os.path.join(self.root, *name.split('.'))
    if os.path.isfile(base + '.py'):
      return base + '.py', False
    if os.path.isfile(os.path.join(base, '__init__.py')):
      return os.path.join(base, '__init__.py'), True
    return None, False

  def find_spec(self, name, path=None, target=None):
    if name not in REAL_IN_STUBS:
      if name in STUBS or name.startswith(tuple(s + '.' for s in STUBS)):
        return importlib.machinery.ModuleSpec(name, tfapi._StubFinder(()), is_package=True)
      if name not in REAL:
        return None
    p, is_pkg = self._path(name)
    if p is None:
      return None
    spec = importlib.machinery.ModuleSpec(name, self, origin=p, is_package=is_pkg)
    if is_pkg:
      spec.submodule_search_locations = [os.path.dirname(p)]
    return spec

  def create_module(self, spec):
    return None

  def exec_module(self, module):
    path = module.__spec__.origin
    with open(path) as fh:
      src = fh.read()
    src3 = _to_py3(src, path, '.' not in module.__name__)
    tree = ast.parse(src3, path)
    future_div = any(isinstance(n, ast.ImportFrom) and n.module == '__future__' and
                     any(a.name == 'division' for a in n.names) for n in tree.body)
    if not future_div:
      tree = ast.fix_missing_locations(_Py2Div().visit(tree))
    module.__dict__['__py2div__'] = py2div
    module.__file__ = path
    sys.dont_write_bytecode = True
    exec(compile(tree, path, 'exec'), module.__dict__)


_installed = {}


def install(root='/root/reference'):
  """Installs the TF shim and the reference importer (idempotent)."""
  if not os.path.isdir(root):
    raise RuntimeError('reference tree %s not found (the shim only runs in the build container)' % root)
  if not _installed:
    _installed['tf'] = tfapi.install()
    sys.meta_path.insert(0, _ReferenceFinder(root))
    _installed['root'] = root
  return _installed['tf']
