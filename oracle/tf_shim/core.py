"""TEST INFRASTRUCTURE -- a minimal eager stand-in for the TensorFlow-1.8 API surface that the reference's hot path
touches, so that the reference's OWN source (nets/pggan.py, nets/pggan_utils.py, libs/*, twingan.py,
image_generation.py under /root/reference) can be executed in this container, where TensorFlow is not installed.

Only oracle/ref_runner.py and tools/make_ref_golden.py use it (here, never on the GPU box: /root/reference does not
exist there); the product never imports it.  What it pins: the reference's composition -- which layers, in which
order, under which variable names, with which constants, reductions and loss weights.  What it does NOT pin: the
TensorFlow primitives themselves (conv2d padding, moments, avg_pool, resize ...), which are restated here from
their documented TF-1.8 semantics on float64 torch tensors.

Every tensor is a float64 (or int64 / bool) CPU torch tensor wrapped in `Tensor`, which carries the DECLARED tf
dtype (so that dtype-dependent constants in the reference pick the float32 branch) and a TF-style static shape.
"""
import contextlib
import re

import numpy as np
import torch

F64 = torch.float64


# ------------------------------------------------------------------------------------------------------------------
# dtypes / shapes
# ------------------------------------------------------------------------------------------------------------------
class DType(object):
  def __init__(self, name, floating):
    self.name, self.is_floating = name, floating

  @property
  def base_dtype(self):
    return self

  @property
  def is_integer(self):
    return not self.is_floating and self.name != 'bool'

  def is_compatible_with(self, other):
    return self == other

  def __eq__(self, other):
    return isinstance(other, DType) and other.name == self.name

  def __ne__(self, other):
    return not self.__eq__(other)

  def __hash__(self):
    return hash(self.name)

  def __repr__(self):
    return 'tf.' + self.name


float16, float32, float64 = DType('float16', True), DType('float32', True), DType('float64', True)
int32, int64, bool_ = DType('int32', False), DType('int64', False), DType('bool', False)
uint8 = DType('uint8', False)


class Dimension(object):
  """tf.Dimension: an int (or None) with Python-2 integer division."""

  def __init__(self, value):
    self.value = value.value if isinstance(value, Dimension) else (None if value is None else int(value))

  def __int__(self):
    return int(self.value)

  __index__ = __int__

  def __eq__(self, other):
    return self.value == (other.value if isinstance(other, Dimension) else other)

  def __ne__(self, other):
    return not self.__eq__(other)

  def __hash__(self):
    return hash(self.value)

  def __lt__(self, other):
    return self.value < int(other)

  def __le__(self, other):
    return self.value <= int(other)

  def __gt__(self, other):
    return self.value > int(other)

  def __ge__(self, other):
    return self.value >= int(other)

  def __mul__(self, other):
    return Dimension(self.value * int(other))

  __rmul__ = __mul__

  def __add__(self, other):
    return Dimension(self.value + int(other))

  __radd__ = __add__

  def __sub__(self, other):
    return Dimension(self.value - int(other))

  def __floordiv__(self, other):
    return Dimension(self.value // int(other))

  __truediv__ = __floordiv__      # tf.Dimension.__div__ is a floor division

  def __float__(self):
    return float(self.value)

  def __repr__(self):
    return 'Dimension(%s)' % self.value

  def __str__(self):
    return '?' if self.value is None else str(self.value)


class TensorShape(object):
  def __init__(self, dims):
    if isinstance(dims, TensorShape):
      dims = dims.dims
    self.dims = None if dims is None else [Dimension(d) for d in dims]

  @property
  def ndims(self):
    return None if self.dims is None else len(self.dims)

  def __len__(self):
    return len(self.dims)

  def __iter__(self):
    return iter(self.dims)

  def __getitem__(self, key):
    if isinstance(key, slice):
      return TensorShape(self.dims[key])
    return self.dims[key]

  def as_list(self):
    return [d.value for d in self.dims]

  def is_fully_defined(self):
    return self.dims is not None and all(d.value is not None for d in self.dims)

  def num_elements(self):
    return int(np.prod(self.as_list()))

  def with_rank(self, r):
    assert self.ndims == r
    return self

  def with_rank_at_least(self, r):
    assert self.ndims >= r
    return self

  def is_compatible_with(self, other):
    return self.as_list() == TensorShape(other).as_list()

  def assert_is_compatible_with(self, other):
    assert self.is_compatible_with(other), (self, other)

  def __eq__(self, other):
    try:
      return self.as_list() == TensorShape(other).as_list()
    except TypeError:
      return False

  def __ne__(self, other):
    return not self.__eq__(other)

  def __repr__(self):
    return 'TensorShape(%s)' % (self.as_list() if self.dims is not None else None)


def shape_list(shape):
  """shape-like (TensorShape / list of ints or Dimensions / Tensor) -> list of python ints (None -> 1)."""
  if isinstance(shape, Tensor):
    return [int(v) for v in shape.t.tolist()]
  if isinstance(shape, TensorShape):
    shape = shape.dims
  out = []
  for d in shape:
    if isinstance(d, Dimension):
      d = d.value
    if isinstance(d, Tensor):
      d = int(d.t.item())
    out.append(1 if d is None else int(d))
  return out


# ------------------------------------------------------------------------------------------------------------------
# tensors
# ------------------------------------------------------------------------------------------------------------------
class _Op(object):
  def __init__(self, name):
    self.name = name
    self.type = 'Shim'


def raw(x):
  """Anything tensor-like -> torch tensor (float64 for floating data)."""
  if isinstance(x, Tensor):
    return x.t
  if isinstance(x, torch.Tensor):
    return x
  if isinstance(x, Dimension):
    return torch.tensor(float(x.value), dtype=F64)
  if isinstance(x, (bool, np.bool_)):
    return torch.tensor(bool(x))
  if isinstance(x, (int, np.integer)):
    return torch.tensor(float(x), dtype=F64)
  if isinstance(x, (float, np.floating)):
    return torch.tensor(float(x), dtype=F64)
  if isinstance(x, (list, tuple)) and any(isinstance(v, Tensor) for v in x):
    return torch.stack([raw(v) for v in x])
  a = np.asarray(x)
  if a.dtype.kind == 'f' or a.dtype.kind in 'iu':
    return torch.tensor(a.astype(np.float64), dtype=F64)
  return torch.tensor(a)


class Tensor(object):
  __array_ufunc__ = None      # numpy scalars defer to __r*__ instead of building object arrays

  def __init__(self, t, dtype=None, name=None):
    assert isinstance(t, torch.Tensor), type(t)
    self.t = t
    if dtype is None:
      dtype = float32 if t.is_floating_point() else (bool_ if t.dtype == torch.bool else int32)
    self._dtype = dtype
    self.name = (name or 'Tensor') + ':0'
    self.op = _Op(name or 'Tensor')
    self.device = ''
    self.graph = None

  @property
  def dtype(self):
    return self._dtype

  @property
  def shape(self):
    return TensorShape(list(self.t.shape))

  def get_shape(self):
    return self.shape

  def set_shape(self, shape):
    pass

  def _bin(self, other, fn, rev=False):
    a, b = self.t, raw(other)
    if rev:
      a, b = b, a
    return Tensor(fn(a, b), self._dtype)

  def __add__(self, o):
    return self._bin(o, torch.add)

  def __radd__(self, o):
    return self._bin(o, torch.add, True)

  def __sub__(self, o):
    return self._bin(o, torch.sub)

  def __rsub__(self, o):
    return self._bin(o, torch.sub, True)

  def __mul__(self, o):
    return self._bin(o, torch.mul)

  def __rmul__(self, o):
    return self._bin(o, torch.mul, True)

  def __truediv__(self, o):
    return self._bin(o, torch.div)

  def __rtruediv__(self, o):
    return self._bin(o, torch.div, True)

  def __floordiv__(self, o):
    return self._bin(o, lambda a, b: torch.floor(a / b))

  def __pow__(self, o):
    return self._bin(o, torch.pow)

  def __neg__(self):
    return Tensor(-self.t, self._dtype)

  def __abs__(self):
    return Tensor(self.t.abs(), self._dtype)

  def __getitem__(self, key):
    return Tensor(self.t[key], self._dtype)

  def __lt__(self, o):
    return Tensor(self.t < raw(o), bool_)

  def __gt__(self, o):
    return Tensor(self.t > raw(o), bool_)

  def __le__(self, o):
    return Tensor(self.t <= raw(o), bool_)

  def __ge__(self, o):
    return Tensor(self.t >= raw(o), bool_)

  def __bool__(self):
    raise TypeError('a tf.Tensor is not a Python bool')

  def __hash__(self):
    return id(self)

  def __eq__(self, o):
    return self is o

  def eval(self):
    return self.t.detach().numpy()

  def numpy(self):
    return self.t.detach().numpy()

  def __repr__(self):
    return '<shim Tensor %s %s %s>' % (self.name, list(self.t.shape), self._dtype)


class Variable(Tensor):
  def __init__(self, full_name, t, dtype, trainable):
    Tensor.__init__(self, t, dtype, full_name)
    self.trainable = trainable
    self.initializer = None

  def value(self):
    return self

  def read_value(self):
    return self

  def initialized_value(self):
    return self

  def assign(self, value):
    with torch.no_grad():
      self.t.copy_(raw(value).reshape(self.t.shape))
    return self


def wrap(t, like=None, dtype=None, name=None):
  if dtype is None and isinstance(like, Tensor):
    dtype = like.dtype
  return Tensor(t, dtype, name)


# ------------------------------------------------------------------------------------------------------------------
# graph state: variables, scopes, collections, flags, random draws
# ------------------------------------------------------------------------------------------------------------------
class _AutoReuse(object):
  def __repr__(self):
    return 'AUTO_REUSE'


AUTO_REUSE = _AutoReuse()


class VariableScope(object):
  def __init__(self, name, reuse):
    self.name, self.reuse = name, reuse
    self.original_name_scope = name + '/' if name else ''
    self.custom_getter = None
    self.dtype = float32
    self.initializer = None

  def reuse_variables(self):
    self.reuse = True

  def set_partitioner(self, partitioner):
    pass


class State(object):
  def __init__(self, seed=0):
    self.reset(seed)

  def reset(self, seed=0):
    self.variables = {}                 # full name -> Variable (creation order preserved)
    self.scope_stack = [VariableScope('', None)]
    self.scope_count = {}               # tf's VariableStore.variable_scopes_count
    self.collections = {}
    self.gen = torch.Generator().manual_seed(seed)
    self.random_log = []                # (op name, torch tensor) of every random op, in call order
    self.aug_log = []                   # (kind, value) of the colour distortions applied to a LIVE image, in order
    self.global_step = None
    self.name_scope = ''
    self.placeholder_batch = 1
    self.opt_count = 0                  # optimizers built in the current graph (tfapi.AdamOptimizer)
    self.placeholder_feed = {}          # placeholder name -> array fed to it (the inference branch; default zeros)
    self.preset = {}                    # full variable name -> numpy value used instead of the initializer
    self.deferred = []                  # assign ops waiting for run_update_ops()
    self.eager_updates = False          # True: assign ops run where they are created (see tfapi._Assign.schedule)


STATE = State()


def current_scope():
  return STATE.scope_stack[-1]


def _unique_scope(prefix):
  """variable_scope.py::_get_unique_variable_scope (TF 1.8)."""
  cur = current_scope().name
  name = cur + '/' + prefix if cur else prefix
  if STATE.scope_count.get(name, 0) == 0:
    return prefix
  idx = 1
  while STATE.scope_count.get(name + '_%d' % idx, 0) > 0:
    idx += 1
  return prefix + '_%d' % idx


@contextlib.contextmanager
def variable_scope(name_or_scope, default_name=None, values=None, initializer=None, regularizer=None, reuse=None,
                   dtype=None, custom_getter=None, **unused):
  parent = current_scope()
  if name_or_scope is None:
    assert default_name is not None
    if reuse:
      raise ValueError('reuse=True cannot be used without a name_or_scope')
    name_or_scope = _unique_scope(default_name)
  if isinstance(name_or_scope, VariableScope):
    full = name_or_scope.name                      # re-entering a captured scope: absolute name
    inherited = name_or_scope.reuse
  else:
    full = parent.name + '/' + name_or_scope if parent.name else name_or_scope
    inherited = None
  # reuse: True / AUTO_REUSE stick; False and None inherit from the enclosing scope (TF 1.8 semantics)
  if reuse is True or reuse is AUTO_REUSE:
    eff = reuse
  else:
    eff = inherited if inherited else parent.reuse
  STATE.scope_count[full] = STATE.scope_count.get(full, 0) + 1
  sc = VariableScope(full, eff)
  # custom getters nest: a scope without its own keeps the enclosing one (or, re-entered, the one it was captured with).
  # Only getters that mark themselves `layer_getter` are honoured (contrib layers._build_variable_getter, libs/sn.py:
  # 199-204); the dtype getter of deployment/model_deploy.py:146-183 is the identity for float32 variables and ignored.
  if custom_getter is not None and getattr(custom_getter, 'layer_getter', False):
    sc.custom_getter = custom_getter
  elif isinstance(name_or_scope, VariableScope) and name_or_scope.custom_getter is not None:
    sc.custom_getter = name_or_scope.custom_getter
  else:
    sc.custom_getter = parent.custom_getter
  STATE.scope_stack.append(sc)
  try:
    yield sc
  finally:
    STATE.scope_stack.pop()
    for k in list(STATE.scope_count):               # VariableStore.close_variable_subscopes
      if k.startswith(full + '/'):
        STATE.scope_count[k] = 0


def get_variable_scope():
  return current_scope()


@contextlib.contextmanager
def name_scope(name=None, default_name=None, values=None):
  """Only what the hot path needs of tf.name_scope: a prefix for the names of loss tensors (so that
  tf.get_collection(collection, clone.scope) can pick one clone's losses, deployment/model_deploy.py:258) and the
  'scope/' string it yields.  A name ending in '/' re-enters that scope verbatim."""
  name = name or default_name or ''
  if name.endswith('/'):
    full = name
  elif name:
    full = STATE.name_scope + name + '/'
  else:
    full = STATE.name_scope
  saved, STATE.name_scope = STATE.name_scope, full
  try:
    yield full
  finally:
    STATE.name_scope = saved


def get_variable(name, shape=None, dtype=None, initializer=None, regularizer=None, trainable=True, collections=None,
                 **unused):
  sc = current_scope()
  if sc.custom_getter is not None and not getattr(STATE, 'in_getter', False):
    STATE.in_getter = True
    try:
      return sc.custom_getter(name, shape, dtype, initializer, regularizer, trainable, collections)
    finally:
      STATE.in_getter = False
  full = sc.name + '/' + name if sc.name else name
  if full in STATE.variables:
    if not sc.reuse:
      raise ValueError('Variable %s already exists, disallowed (reuse=%s)' % (full, sc.reuse))
    return STATE.variables[full]
  if sc.reuse is True:
    raise ValueError('Variable %s does not exist, or was not created with tf.get_variable() (reuse=True)' % full)
  dtype = dtype or float32
  if full in STATE.preset:
    t = torch.tensor(np.asarray(STATE.preset[full], dtype=np.float64), dtype=F64)
    if shape is not None:
      assert list(t.shape) == shape_list(shape), (full, list(t.shape), shape_list(shape))
  else:
    if initializer is None:
      initializer = glorot_uniform_initializer()
    if getattr(initializer, '_tf_initializer_class', False):
      # the CLASS passed uninstantiated (initializer=tf.zeros_initializer, image_generation.py:1035-1038): TensorFlow
      # instantiates it with the variable's dtype (variable_scope._get_single_variable)
      initializer = initializer(dtype=dtype)
    if callable(initializer):
      t = raw(initializer(shape_list(shape), dtype=dtype))
    else:
      t = raw(initializer).clone()
  t = t.detach().clone()
  if dtype.is_floating:
    t = t.to(F64)
    t.requires_grad_(bool(trainable))
  v = Variable(full, t, dtype, trainable)
  STATE.variables[full] = v
  if isinstance(collections, str):      # tf.Variable itself rejects a bare string; nothing on the path gets here with one
    collections = [collections]
  keys = list(collections) if collections else ['variables']
  if 'variables' not in keys:
    keys.append('variables')
  if trainable:
    keys.append('trainable_variables')
  for k in dict.fromkeys(keys):
    add_to_collection(k, v)
  return v


def add_to_collection(name, value):
  STATE.collections.setdefault(name, []).append(value)


def add_to_collections(names, value):
  if isinstance(names, str):
    names = [names]
  for n in names or []:
    add_to_collection(n, value)


def get_collection(name, scope=None):
  items = list(STATE.collections.get(name, []))
  if scope:
    items = [i for i in items if hasattr(i, 'name') and re.match(scope, i.name)]
  return items


def get_collection_ref(name):
  return STATE.collections.setdefault(name, [])


class GraphKeys(object):
  GLOBAL_VARIABLES = 'variables'
  VARIABLES = 'variables'
  TRAINABLE_VARIABLES = 'trainable_variables'
  MODEL_VARIABLES = 'model_variables'
  UPDATE_OPS = 'update_ops'
  LOSSES = 'losses'
  REGULARIZATION_LOSSES = 'regularization_losses'
  SUMMARIES = 'summaries'
  MOVING_AVERAGE_VARIABLES = 'moving_average_variables'
  GLOBAL_STEP = 'global_step'


# ---- initializers ---------------------------------------------------------------------------------------------
def zeros_initializer(dtype=None):
  return lambda shape, dtype=None, partition_info=None: Tensor(torch.zeros(shape_list(shape), dtype=F64))


def ones_initializer(dtype=None):
  return lambda shape, dtype=None, partition_info=None: Tensor(torch.ones(shape_list(shape), dtype=F64))


zeros_initializer._tf_initializer_class = True      # classes in TensorFlow: get_variable accepts them uninstantiated
ones_initializer._tf_initializer_class = True


def constant_initializer(value=0.0, dtype=None):
  def init(shape, dtype=None, partition_info=None):
    v = raw(value)
    return Tensor(v.expand(shape_list(shape)).clone() if v.dim() == 0 else v.reshape(shape_list(shape)).clone())
  return init


def random_normal_initializer(mean=0.0, stddev=1.0, seed=None, dtype=None):
  def init(shape, dtype=None, partition_info=None):
    return Tensor(mean + stddev * torch.randn(shape_list(shape), dtype=F64, generator=STATE.gen))
  return init


def truncated_normal_initializer(mean=0.0, stddev=1.0, seed=None, dtype=None):
  def init(shape, dtype=None, partition_info=None):
    x = torch.randn(shape_list(shape), dtype=F64, generator=STATE.gen)
    for _ in range(8):      # redraw the |x| > 2 tail, as tf.truncated_normal does
      bad = x.abs() > 2
      if not bad.any():
        break
      x = torch.where(bad, torch.randn(x.shape, dtype=F64, generator=STATE.gen), x)
    return Tensor(mean + stddev * x.clamp(-2, 2))
  return init


def glorot_uniform_initializer(seed=None, dtype=None):
  def init(shape, dtype=None, partition_info=None):
    shape = shape_list(shape)
    rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
    fan_in = shape[-2] * rf if len(shape) >= 2 else shape[0]
    fan_out = shape[-1] * rf
    lim = float(np.sqrt(6.0 / (fan_in + fan_out)))
    return Tensor((torch.rand(shape, dtype=F64, generator=STATE.gen) * 2 - 1) * lim)
  return init


# ---- flags --------------------------------------------------------------------------------------------------------
class _Flags(object):
  def __init__(self):
    object.__setattr__(self, '_v', {})

  def __getattr__(self, k):
    try:
      return self._v[k]
    except KeyError:
      raise AttributeError('flag %s not defined' % k)

  def __setattr__(self, k, v):
    self._v[k] = v

  def __contains__(self, k):
    return k in self._v

  def flag_values_dict(self):
    return dict(self._v)


FLAGS = _Flags()


def _define(name, default, help=None, **unused):
  if name not in FLAGS:
    setattr(FLAGS, name, default)


class flags(object):
  FLAGS = FLAGS
  DEFINE_string = staticmethod(_define)
  DEFINE_boolean = staticmethod(_define)
  DEFINE_bool = staticmethod(_define)
  DEFINE_integer = staticmethod(_define)
  DEFINE_float = staticmethod(_define)
  DEFINE_enum = staticmethod(lambda name, default, enum_values=None, help=None, **kw: _define(name, default))
  DEFINE_list = staticmethod(_define)
  DEFINE_multi_integer = staticmethod(_define)
  DEFINE_multi_string = staticmethod(_define)
