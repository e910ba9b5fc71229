"""TEST INFRASTRUCTURE -- the `tensorflow` module tree of the shim (see core.py for what this is and is not).

Each function restates the documented TF-1.8 behaviour of the op of the same name on float64 torch tensors; the
non-obvious ones say which TF behaviour they follow."""
import contextlib
import functools
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

from . import core
from .core import (F64, STATE, Tensor, Variable, TensorShape, Dimension, raw, wrap, shape_list, float32, float16,
                   float64, int32, int64, bool_)


# ------------------------------------------------------------------------------------------------------------------
# plain ops
# ------------------------------------------------------------------------------------------------------------------
def _axes(axis, ndim):
  if axis is None:
    return list(range(ndim))
  if isinstance(axis, (int, np.integer, Dimension)):
    axis = [int(axis)]
  return [int(a) % ndim for a in axis]


def constant(value, dtype=None, shape=None, name=None, verify_shape=False):
  t = raw(value)
  if dtype is not None and not dtype.is_floating and t.is_floating_point():
    t = t.to(torch.int64)
  if shape is not None:
    t = t.expand(shape_list(shape)).clone() if t.dim() == 0 else t.reshape(shape_list(shape))
  return Tensor(t, dtype, name)


def convert_to_tensor(value, dtype=None, name=None, preferred_dtype=None):
  if isinstance(value, Tensor):
    return value
  return constant(value, dtype=dtype, name=name)


def cast(x, dtype, name=None):
  t = raw(x)
  if dtype.is_floating:
    t = t.to(F64)
  elif dtype == bool_:
    t = t != 0
  else:
    t = t.to(torch.int64) if not t.is_floating_point() else torch.trunc(t).to(torch.int64)
  return Tensor(t, dtype, name)


def to_float(x, name=None):
  return cast(x, float32)


def identity(x, name=None):
  x = convert_to_tensor(x)
  return Tensor(x.t, x.dtype, name)


def stop_gradient(x, name=None):
  return wrap(raw(x).detach(), x)


def reshape(x, shape, name=None):
  return wrap(raw(x).reshape(shape_list(shape)), x)


def expand_dims(x, axis=None, name=None, dim=None):
  axis = dim if axis is None else axis
  return wrap(raw(x).unsqueeze(int(axis)), x)


def squeeze(x, axis=None, name=None, squeeze_dims=None):
  axis = squeeze_dims if axis is None else axis
  t = raw(x)
  if axis is None:
    return wrap(t.squeeze(), x)
  for a in sorted(_axes(axis, t.dim()), reverse=True):
    assert t.shape[a] == 1
    t = t.squeeze(a)
  return wrap(t, x)


def concat(values, axis, name=None):
  return wrap(torch.cat([raw(v) for v in values], dim=int(axis)), values[0])


def stack(values, axis=0, name=None):
  return wrap(torch.stack([raw(v) for v in values], dim=int(axis)), values[0] if isinstance(values[0], Tensor) else None)


def transpose(x, perm=None, name=None):
  t = raw(x)
  return wrap(t.permute(*[int(p) for p in perm]) if perm is not None else t.t(), x)


def tile(x, multiples, name=None):
  return wrap(raw(x).repeat(*shape_list(multiples)), x)


def pad(x, paddings, mode='CONSTANT', name=None, constant_values=0):
  assert mode == 'CONSTANT'
  flat = []
  for lo, hi in reversed([tuple(p) for p in paddings]):
    flat += [int(lo), int(hi)]
  return wrap(F.pad(raw(x), flat, value=float(constant_values)), x)


def shape(x, name=None, out_type=None):
  return Tensor(torch.tensor(list(raw(x).shape), dtype=torch.int64), int32)


def zeros_like(x, dtype=None, name=None):
  return wrap(torch.zeros_like(raw(x)), x)


def ones_like(x, dtype=None, name=None):
  return wrap(torch.ones_like(raw(x)), x)


def zeros(shape, dtype=float32, name=None):
  return Tensor(torch.zeros(shape_list(shape), dtype=F64), dtype)


def ones(shape, dtype=float32, name=None):
  return Tensor(torch.ones(shape_list(shape), dtype=F64), dtype)


def _unary(fn):
  def op(x, name=None):
    return wrap(fn(raw(x)), x, name=name)
  return op


def _binary(fn):
  def op(x, y, name=None):
    like = x if isinstance(x, Tensor) else y
    return wrap(fn(raw(x), raw(y)), like, name=name)
  return op


sqrt, rsqrt, square, exp, log = map(_unary, (torch.sqrt, torch.rsqrt, torch.square, torch.exp, torch.log))
negative, tanh, sigmoid, abs_ = map(_unary, (torch.neg, torch.tanh, torch.sigmoid, torch.abs))
add, subtract, multiply, divide = map(_binary, (torch.add, torch.sub, torch.mul, torch.div))
minimum, maximum, pow_ = map(_binary, (torch.minimum, torch.maximum, torch.pow))


def add_n(inputs, name=None):
  out = raw(inputs[0])
  for v in inputs[1:]:
    out = out + raw(v)
  return wrap(out, inputs[0], name=name)


def clip_by_value(x, lo, hi, name=None):
  if x is None:      # a dead tensor of a control_flow_ops.switch stays dead
    return None
  return wrap(torch.minimum(torch.maximum(raw(x), raw(lo)), raw(hi)), x)


def where(cond, x=None, y=None, name=None):
  return wrap(torch.where(raw(cond), raw(x), raw(y)), x)


def _compare(fn):
  def op(x, y, name=None):
    return Tensor(fn(raw(x), raw(y)), bool_)
  return op


equal, not_equal, greater, less = map(_compare, (torch.eq, torch.ne, torch.gt, torch.lt))
greater_equal, less_equal = map(_compare, (torch.ge, torch.le))


def _reduce(fn):
  def op(x, axis=None, keepdims=None, name=None, reduction_indices=None, keep_dims=None):
    axis = reduction_indices if axis is None else axis
    keep = bool(keepdims if keepdims is not None else keep_dims)
    t = raw(x)
    if t.dim() == 0:
      return wrap(t, x, name=name)
    return wrap(fn(t, dim=_axes(axis, t.dim()), keepdim=keep), x, name=name)
  return op


reduce_mean = _reduce(torch.mean)
reduce_sum = _reduce(torch.sum)
reduce_max = _reduce(torch.amax)
reduce_min = _reduce(torch.amin)


def matmul(a, b, transpose_a=False, transpose_b=False, name=None):
  x, y = raw(a), raw(b)
  if transpose_a:
    x = x.transpose(-1, -2)
  if transpose_b:
    y = y.transpose(-1, -2)
  return wrap(torch.matmul(x, y), a)


def tensordot(a, b, axes, name=None):
  if isinstance(axes, (int, np.integer)):
    return wrap(torch.tensordot(raw(a), raw(b), dims=int(axes)), a)
  return wrap(torch.tensordot(raw(a), raw(b), dims=([list(axes[0]), list(axes[1])])), a)


# ---- random ops: every draw is logged so that the oracle can be fed the same numbers -----------------------------
def _draw(kind, shape, name, scale, shift):
  shape = shape_list(shape)
  if kind == 'normal':
    t = torch.randn(shape, dtype=F64, generator=STATE.gen)
  else:
    t = torch.rand(shape, dtype=F64, generator=STATE.gen)
  t = shift + scale * t
  STATE.random_log.append((name or kind, t))      # the value the graph sees
  return t


def random_normal(shape, mean=0.0, stddev=1.0, dtype=float32, seed=None, name=None):
  return Tensor(_draw('normal', shape, name, stddev, mean), dtype, name)


def random_uniform(shape, minval=0, maxval=None, dtype=float32, seed=None, name=None):
  maxval = 1.0 if maxval is None else maxval
  t = _draw('uniform', shape, name, maxval - minval, minval)
  if not dtype.is_floating:      # integer draws: uniform over [minval, maxval)
    t = torch.floor(t)
    STATE.random_log[-1] = (STATE.random_log[-1][0], t)
  return Tensor(t, dtype, name)


# ---- control flow / state ------------------------------------------------------------------------------------
@contextlib.contextmanager
def control_dependencies(deps):
  yield


@contextlib.contextmanager
def _null_context(*a, **k):
  yield


class _Assign(Tensor):
  """A deferred assignment (graph-mode `tf.assign` only runs when fetched): executed by run_update_ops()."""

  def __init__(self, var, value_t):
    Tensor.__init__(self, value_t, var.dtype, 'assign')
    self.var = var

  def schedule(self):
    """Deferred to run_update_ops() -- or, with STATE.eager_updates, executed now: program order is one of the
    schedules a TF executor may pick for update ops that have no dependencies between them."""
    if getattr(STATE, 'eager_updates', False):
      self.run()
    else:
      STATE.deferred.append(self)
    return self

  def run(self):
    with torch.no_grad():
      self.var.t.copy_(self.t.detach().reshape(self.var.t.shape))


class _MovingAverageAssign(_Assign):
  """moving_averages.assign_moving_average: `variable -= (variable - value) * (1 - decay)`.  Its in-graph value is
  computed from the variable as it is now; when it RUNS (run_update_ops) it reads the variable again, so that several
  updates of one variable in a run compose sequentially, as separately scheduled assign_sub ops do."""

  def __init__(self, var, value, decay):
    self.value, self.decay = raw(value).detach(), float(decay)
    _Assign.__init__(self, var, var.t.detach() - (var.t.detach() - self.value) * (1.0 - self.decay))

  def run(self):
    with torch.no_grad():
      self.var.t.sub_((self.var.t - self.value.reshape(self.var.t.shape)) * (1.0 - self.decay))


def assign_moving_average(variable, value, decay, zero_debias=True, name=None):
  assert not zero_debias, 'tf shim: zero_debias moving averages are not implemented'
  return _MovingAverageAssign(variable, value, decay).schedule()


def assign(ref, value, validate_shape=None, use_locking=None, name=None):
  return _Assign(ref, raw(value).detach().clone()).schedule()


def assign_add(ref, value, use_locking=None, name=None):
  return assign(ref, ref.t.detach() + raw(value))


def assign_sub(ref, value, use_locking=None, name=None):
  return assign(ref, ref.t.detach() - raw(value))


def run_update_ops(ops=None):
  """Executes the deferred assignments (all of them, or the ones reachable from `ops`) exactly once."""
  todo = STATE.deferred if ops is None else [o for o in ops if isinstance(o, _Assign)]
  for a in todo:
    a.run()
  STATE.deferred = [a for a in STATE.deferred if a not in todo]


def group(*inputs, **kw):
  return list(inputs)


def no_op(name=None):
  return None


def cond(pred, true_fn=None, false_fn=None, name=None, fn1=None, fn2=None, strict=False):
  true_fn, false_fn = true_fn or fn1, false_fn or fn2
  p = bool(raw(pred).item()) if isinstance(pred, Tensor) else bool(pred)
  return true_fn() if p else false_fn()


def while_loop(cond_fn, body, loop_vars, **unused):
  loop_vars = tuple(loop_vars)
  while True:
    c = cond_fn(*loop_vars)
    c = bool(raw(c).item()) if isinstance(c, Tensor) else bool(c)
    if not c:
      return loop_vars
    out = body(*loop_vars)
    loop_vars = tuple(out) if isinstance(out, (tuple, list)) else (out,)


def gradients(ys, xs, grad_ys=None, name=None, **unused):
  ys = ys if isinstance(ys, (list, tuple)) else [ys]
  xs_l = xs if isinstance(xs, (list, tuple)) else [xs]
  total = sum(raw(y).sum() for y in ys)
  gs = torch.autograd.grad(total, [raw(x) for x in xs_l], create_graph=True, allow_unused=True)
  return [None if g is None else wrap(g, x) for g, x in zip(gs, xs_l)]


def placeholder(dtype, shape=None, name=None):
  """Zeros; an unknown (None) dimension takes STATE.placeholder_batch, so that the inference branch of the graph
  (twingan.py:290-363) is built with the same batch as the training tensors it is mixed with."""
  if name in STATE.placeholder_feed:      # a "feed_dict" for the inference branch
    return Tensor(torch.tensor(np.asarray(STATE.placeholder_feed[name], np.float64)), dtype, name)
  dims = [STATE.placeholder_batch if (d.value if isinstance(d, Dimension) else d) is None else int(d)
          for d in (shape.dims if isinstance(shape, TensorShape) else shape)]
  return Tensor(torch.zeros(dims, dtype=F64), dtype, name)


def placeholder_with_default(input, shape, name=None):
  return convert_to_tensor(input)


# ---- nn ------------------------------------------------------------------------------------------------------
def _same_pad(n, k, s):
  out = -(-n // s)
  total = max((out - 1) * s + k - n, 0)
  return total // 2, total - total // 2


def nn_conv2d(input, filter, strides, padding, use_cudnn_on_gpu=True, data_format='NHWC', dilations=None, name=None):
  """tf.nn.conv2d, NHWC x HWIO; 'SAME' pads (total // 2) before and the rest after, as TF does."""
  x = raw(input).permute(0, 3, 1, 2)
  w = raw(filter).permute(3, 2, 0, 1)
  sh, sw = int(strides[1]), int(strides[2])
  if padding == 'SAME':
    pt, pb = _same_pad(x.shape[2], w.shape[2], sh)
    pl, pr = _same_pad(x.shape[3], w.shape[3], sw)
    x = F.pad(x, [pl, pr, pt, pb])
  else:
    assert padding == 'VALID', padding
  return wrap(F.conv2d(x, w, stride=(sh, sw)).permute(0, 2, 3, 1), input)


def nn_avg_pool(value, ksize, strides, padding, data_format='NHWC', name=None):
  assert padding == 'VALID'
  x = raw(value).permute(0, 3, 1, 2)
  y = F.avg_pool2d(x, (int(ksize[1]), int(ksize[2])), (int(strides[1]), int(strides[2])))
  return wrap(y.permute(0, 2, 3, 1), value)


def nn_bias_add(value, bias, data_format=None, name=None):
  return wrap(raw(value) + raw(bias), value)


def nn_moments(x, axes, shift=None, name=None, keep_dims=False, keepdims=None):
  keep = bool(keepdims if keepdims is not None else keep_dims)
  t = raw(x)
  ax = _axes(axes, t.dim())
  mean = t.mean(dim=ax, keepdim=True)
  var = ((t - mean) ** 2).mean(dim=ax, keepdim=True)      # population variance of the centred values
  if not keep:
    for a in sorted(ax, reverse=True):
      mean, var = mean.squeeze(a), var.squeeze(a)
  return wrap(mean, x), wrap(var, x)


def nn_batch_normalization(x, mean, variance, offset, scale, variance_epsilon, name=None):
  inv = torch.rsqrt(raw(variance) + raw(variance_epsilon))
  if scale is not None:
    inv = inv * raw(scale)
  shift = -raw(mean) * inv
  if offset is not None:
    shift = raw(offset) + shift
  return wrap(raw(x) * inv + shift, x)


def nn_l2_normalize(x, axis=None, epsilon=1e-12, name=None, dim=None):
  axis = dim if axis is None else axis
  t = raw(x)
  ss = (t * t).sum(dim=_axes(axis, t.dim()), keepdim=True)
  return wrap(t * torch.rsqrt(torch.clamp(ss, min=epsilon)), x)


def nn_softmax(logits, axis=-1, name=None, dim=None):
  axis = dim if dim is not None else axis
  return wrap(torch.softmax(raw(logits), dim=int(axis)), logits)


def nn_sigmoid_xent(_sentinel=None, labels=None, logits=None, name=None):
  x, z = raw(logits), raw(labels)
  return wrap(torch.clamp(x, min=0) - x * z + torch.log1p(torch.exp(-x.abs())), logits)


def nn_leaky_relu(features, alpha=0.2, name=None):
  t = raw(features)
  return wrap(torch.maximum(alpha * t, t), features)


def image_resize_nearest(images, size, align_corners=False, name=None):
  """tf.image.resize_nearest_neighbor (align_corners=False): src index = floor(dst * in / out)."""
  t = raw(images)
  oh, ow = shape_list(size)
  ih, iw = t.shape[1], t.shape[2]
  ys = torch.clamp(torch.floor(torch.arange(oh, dtype=F64) * (ih / oh)).long(), max=ih - 1)
  xs = torch.clamp(torch.floor(torch.arange(ow, dtype=F64) * (iw / ow)).long(), max=iw - 1)
  return wrap(t[:, ys][:, :, xs], images)


def image_resize_bilinear(images, size, align_corners=False, name=None):
  """tf.image.resize_bilinear (TF1, align_corners=False): src = dst * in / out, no half-pixel offset."""
  t = raw(images)
  oh, ow = shape_list(size)
  ih, iw = t.shape[1], t.shape[2]

  def taps(o, i):
    # core/kernels/resize_bilinear_op.cc compute_interpolation_weights: `const float in = i * scale; lower = (int64)in;
    # lerp = in - lower` -- scale, product and weight are float32 (visible as ~1e-5 on the weights once in / out is not a
    # dyadic number, e.g. the 320-pixel intermediate of --do_random_cropping at 256)
    src = torch.arange(o, dtype=torch.float32) * torch.tensor(np.float32(i) / np.float32(o), dtype=torch.float32)
    lo = torch.clamp(torch.floor(src).long(), max=i - 1)
    hi = torch.clamp(lo + 1, max=i - 1)
    return lo, hi, (src - lo.to(torch.float32)).to(F64)
  y0, y1, fy = taps(oh, ih)
  x0, x1, fx = taps(ow, iw)
  fy = fy.view(1, oh, 1, 1)
  fx = fx.view(1, 1, ow, 1)
  top = t[:, y0][:, :, x0] * (1 - fx) + t[:, y0][:, :, x1] * fx
  bot = t[:, y1][:, :, x0] * (1 - fx) + t[:, y1][:, :, x1] * fx
  return wrap(top * (1 - fy) + bot * fy, images)


# ---- tf.image ops of the input pipeline (preprocessing/*.py): single images [h, w, c] ---------------------------
class ResizeMethod(object):
  BILINEAR, NEAREST_NEIGHBOR, BICUBIC, AREA = 0, 1, 2, 3


def _int(v):
  return int(raw(v).item()) if isinstance(v, (Tensor, torch.Tensor)) else int(v)


def image_convert_image_dtype(image, dtype, saturate=False, name=None):
  """uint8 -> float: cast * (1 / 255) in float32 (python/ops/image_ops_impl.py convert_image_dtype)."""
  if image.dtype == dtype:
    return image
  assert image.dtype == core.uint8 and dtype.is_floating, 'only uint8 -> float is modelled'
  t = (raw(image).to(torch.float32) * torch.tensor(1.0 / 255.0, dtype=torch.float32)).to(F64)
  return Tensor(t, dtype, name)


def image_pad_to_bounding_box(image, offset_height, offset_width, target_height, target_width):
  t = raw(image)
  oh, ow, th, tw = _int(offset_height), _int(offset_width), _int(target_height), _int(target_width)
  out = torch.zeros((th, tw, t.shape[2]), dtype=t.dtype)
  out[oh:oh + t.shape[0], ow:ow + t.shape[1]] = t
  return wrap(out, image)


def image_crop_to_bounding_box(image, offset_height, offset_width, target_height, target_width):
  oh, ow, th, tw = _int(offset_height), _int(offset_width), _int(target_height), _int(target_width)
  return wrap(raw(image)[oh:oh + th, ow:ow + tw], image)


def image_resize_images(images, size, method=ResizeMethod.BILINEAR, align_corners=False):
  assert method == ResizeMethod.BILINEAR and not align_corners, 'only the bilinear kernel is modelled'
  t = raw(images)
  if t.dim() == 3:
    out = raw(image_resize_bilinear(Tensor(t.unsqueeze(0), images.dtype), size))[0]
    return wrap(out, images)
  return image_resize_bilinear(images, size)


def _rgb_to_hsv(rgb):      # core/kernels/colorspace_op.h
  r, g, b = rgb[..., 0], rgb[..., 1], rgb[..., 2]
  v = rgb.max(dim=-1).values
  rng = v - rgb.min(dim=-1).values
  s = torch.where(v > 0, rng / torch.where(v > 0, v, torch.ones_like(v)), torch.zeros_like(v))
  norm = 1.0 / (6.0 * torch.where(rng > 0, rng, torch.ones_like(rng)))
  h = torch.where(r == v, norm * (g - b), torch.where(g == v, norm * (b - r) + 2.0 / 6.0, norm * (r - g) + 4.0 / 6.0))
  h = torch.where(rng > 0, h, torch.zeros_like(h))
  h = torch.where(h < 0, h + 1.0, h)
  return h, s, v


def _hsv_to_rgb(h, s, v):
  c = s * v
  m = v - c
  dh = h * 6.0
  fm = dh - 2.0 * torch.floor(dh / 2.0)
  x = c * (1 - (fm - 1).abs())
  cat = torch.clamp(torch.floor(dh).long(), 0, 5)
  z = torch.zeros_like(c)
  table = [(c, x, z), (x, c, z), (z, c, x), (z, x, c), (x, z, c), (c, z, x)]
  out = torch.stack([sum(torch.where(cat == k, t[i], z) for k, t in enumerate(table)) for i in range(3)], dim=-1)
  return out + m.unsqueeze(-1)


def image_adjust_saturation(image, saturation_factor, name=None):
  h, s, v = _rgb_to_hsv(raw(image))
  s = torch.clamp(s * raw(saturation_factor), 0.0, 1.0)
  return wrap(_hsv_to_rgb(h, s, v), image)


def image_random_brightness(image, max_delta, seed=None):
  """adjust_brightness(image, U[-max_delta, max_delta)); the draw happens whether or not the branch is live."""
  delta = random_uniform([], -max_delta, max_delta)
  if image is None:
    return None
  STATE.aug_log.append(('brightness', float(raw(delta))))
  return wrap(raw(image) + raw(delta), image)


def image_random_saturation(image, lower, upper, seed=None):
  factor = random_uniform([], lower, upper)
  if image is None:
    return None
  STATE.aug_log.append(('saturation', float(raw(factor))))
  return image_adjust_saturation(image, factor)


def reverse(tensor, axis, name=None):
  return wrap(torch.flip(raw(tensor), dims=[int(a) for a in axis]), tensor)


def matrix_transpose(a, name='matrix_transpose', conjugate=False):
  return wrap(raw(a).transpose(-1, -2), a)


def random_crop(value, size, seed=None, name=None):
  """tf.random_crop (python/ops/random_ops.py): offset = random_uniform(shape(shape), dtype=size.dtype, maxval=size.dtype.max)
  % (shape - size + 1), then tf.slice(value, offset, size).  The offsets and the crop size are logged (aug_log 'crop' =
  (oy, ox, h, w)) so that a restatement can be fed the same rectangle."""
  t = raw(value)
  sz = [int(raw(v).item()) if isinstance(v, Tensor) else int(v) for v in size]
  shape_ = list(t.shape)
  assert len(sz) == len(shape_) and all(a >= b for a, b in zip(shape_, sz)), 'Need value.shape >= size'
  draw = random_uniform([len(shape_)], maxval=2 ** 31 - 1, dtype=int32, name=name or 'random_crop')
  off = [int(d) % (a - b + 1) for d, a, b in zip(raw(draw).tolist(), shape_, sz)]
  STATE.aug_log.append(('crop', (off[0], off[1], sz[0], sz[1])))
  return wrap(t[tuple(slice(o, o + n) for o, n in zip(off, sz))], value)


def cf_switch(data, pred, dtype=None, name=None):
  """control_flow_ops.switch: (output_false, output_true); the branch not taken is dead (None here)."""
  p = bool(raw(pred).item())
  return (None, data) if p else (data, None)


def cf_merge(inputs, name=None):
  live = [(i, v) for i, v in enumerate(inputs) if v is not None]
  assert len(live) == 1, 'merge expects exactly one live input'
  return live[0][1], live[0][0]


# ---- tf.losses ---------------------------------------------------------------------------------------------------
class Reduction(object):
  NONE, SUM, MEAN = 'none', 'weighted_sum', 'weighted_mean'
  SUM_OVER_BATCH_SIZE, SUM_OVER_NONZERO_WEIGHTS = 'weighted_sum_over_batch_size', 'weighted_sum_by_nonzero_weights'
  SUM_BY_NONZERO_WEIGHTS = 'weighted_sum_by_nonzero_weights'


def compute_weighted_loss(losses, weights=1.0, scope=None, loss_collection='losses',
                          reduction=Reduction.SUM_BY_NONZERO_WEIGHTS):
  """tf.losses.compute_weighted_loss: sum(losses * weights) / #(elements whose weight is not 0), 0 if there are none."""
  assert reduction == Reduction.SUM_BY_NONZERO_WEIGHTS
  l = raw(losses)
  w = raw(weights) * torch.ones_like(l)
  present = (w != 0).to(F64).sum()
  total = (l * w).sum()
  loss = torch.where(present > 0, total / torch.clamp(present, min=1.0), torch.zeros_like(total))
  out = Tensor(loss, float32, STATE.name_scope + (scope or 'weighted_loss') + '/value')
  if loss_collection:
    core.add_to_collection(loss_collection, out)
  return out


def absolute_difference(labels, predictions, weights=1.0, scope=None, loss_collection='losses',
                        reduction=Reduction.SUM_BY_NONZERO_WEIGHTS):
  losses = (raw(predictions) - raw(labels)).abs()
  return compute_weighted_loss(Tensor(losses), weights, scope or 'absolute_difference', loss_collection, reduction)


def sigmoid_cross_entropy(multi_class_labels, logits, weights=1.0, label_smoothing=0, scope=None,
                          loss_collection='losses', reduction=Reduction.SUM_BY_NONZERO_WEIGHTS):
  assert not label_smoothing
  losses = nn_sigmoid_xent(labels=multi_class_labels, logits=logits)
  return compute_weighted_loss(losses, weights, scope or 'sigmoid_cross_entropy_loss', loss_collection, reduction)


def cosine_distance(labels, predictions, axis=None, weights=1.0, scope=None, loss_collection='losses',
                    reduction=Reduction.SUM_BY_NONZERO_WEIGHTS, dim=None):
  axis = dim if axis is None else axis
  losses = 1 - (raw(predictions) * raw(labels)).sum(dim=int(axis), keepdim=True)
  return compute_weighted_loss(Tensor(losses), weights, scope or 'cosine_distance_loss', loss_collection, reduction)


def get_losses(scope=None, loss_collection='losses'):
  return core.get_collection(loss_collection, scope)


# ---- tf.train ----------------------------------------------------------------------------------------------------
def get_global_step(graph=None):
  return STATE.global_step


def get_or_create_global_step(graph=None):
  if STATE.global_step is None:
    STATE.global_step = Variable('global_step', torch.zeros((), dtype=torch.int64), int64, False)
  return STATE.global_step


def piecewise_constant(x, boundaries, values, name=None):
  """values[i] for the first i with x <= boundaries[i], else values[-1]."""
  xv = float(raw(x).item())
  for b, v in zip(boundaries, values):
    if xv <= b:
      return Tensor(torch.tensor(float(v), dtype=F64), float32, name)
  return Tensor(torch.tensor(float(values[-1]), dtype=F64), float32, name)


class Optimizer(object):
  """tf.train.Optimizer as the hot path uses it: compute_gradients (deployment/model_deploy.py:302-306) and
  apply_gradients(grads_and_vars, global_step) (image_generation.py:550).  The base class is plain gradient descent."""

  def __init__(self, learning_rate=0.0, *args, **kwargs):
    self._lr = learning_rate

  def compute_gradients(self, loss, var_list=None, **unused):
    var_list = list(var_list) if var_list is not None else core.get_collection(core.GraphKeys.TRAINABLE_VARIABLES)
    gs = torch.autograd.grad(raw(loss), [v.t for v in var_list], allow_unused=True, retain_graph=True)
    return [(None if g is None else Tensor(g, v.dtype, v.op.name + '/grad'), v) for g, v in zip(gs, var_list)]

  def _slot(self, var, name, init):
    """Optimizer state lives in the variable store (reference names: '<var>/Adam', '<var>/Adam_1', 'beta1_power',
    'beta2_power'), so it persists when the graph is rebuilt for the next run."""
    full = (var.op.name + '/' + name) if var is not None else name
    if full not in STATE.variables:
      STATE.variables[full] = Variable(full, init.detach().clone(), float32, False)
    return STATE.variables[full]

  def _apply(self, grad, var):
    with torch.no_grad():
      var.t.sub_(float(raw(self._lr)) * raw(grad).detach())

  def _finish(self):
    pass

  def apply_gradients(self, grads_and_vars, global_step=None, name=None):
    """Runs where it is created (program order): the update of every variable, the optimizer's own state, then
    global_step += 1 -- the order tf.train.Optimizer.apply_gradients enforces with control dependencies."""
    for g, v in grads_and_vars:
      if g is not None:
        self._apply(g, v)
    self._finish()
    if global_step is not None:
      with torch.no_grad():
        global_step.t.add_(1)
    return Tensor(torch.tensor(True), bool_, name or 'apply_gradients')


class AdamOptimizer(Optimizer):
  """tf.train.AdamOptimizer (TF 1.8, training/adam.py): per variable m <- b1 m + (1-b1) g, v <- b2 v + (1-b2) g^2,
  var <- var - lr_t m / (sqrt(v) + eps) with lr_t = lr sqrt(1 - b2^t) / (1 - b1^t); the powers b1^t, b2^t are ONE pair
  of non-slot variables per optimizer object, multiplied by b1, b2 after each apply_gradients call."""

  def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8, use_locking=False, name='Adam'):
    Optimizer.__init__(self, learning_rate)
    self._b1, self._b2, self._eps = float(beta1), float(beta2), float(epsilon)
    # the non-slot power accumulators belong to the optimizer OBJECT (optimizer.py _non_slot_dict); a graph names the
    # second optimizer's 'beta1_power_1' ... -- the k-th Adam built in a run (STATE.opt_count, reset per graph build)
    self._idx = getattr(STATE, 'opt_count', 0)
    STATE.opt_count = self._idx + 1

  def _powers(self):
    sfx = '' if self._idx == 0 else '_%d' % self._idx
    return (self._slot(None, 'beta1_power' + sfx, torch.tensor(self._b1, dtype=F64)),
            self._slot(None, 'beta2_power' + sfx, torch.tensor(self._b2, dtype=F64)))

  def _apply(self, grad, var):
    b1p, b2p = self._powers()
    m = self._slot(var, 'Adam', torch.zeros_like(var.t))
    v = self._slot(var, 'Adam_1', torch.zeros_like(var.t))
    g = raw(grad).detach()
    with torch.no_grad():
      lr_t = float(raw(self._lr)) * torch.sqrt(1 - b2p.t) / (1 - b1p.t)
      m.t.mul_(self._b1).add_((1 - self._b1) * g)
      v.t.mul_(self._b2).add_((1 - self._b2) * g * g)
      var.t.sub_(lr_t * m.t / (torch.sqrt(v.t) + self._eps))

  def _finish(self):
    b1p, b2p = self._powers()
    with torch.no_grad():
      b1p.t.mul_(self._b1)
      b2p.t.mul_(self._b2)


def floormod(x, y, name=None):
  return wrap(torch.remainder(raw(x), raw(y)), x if isinstance(x, Tensor) else y)


def global_norm(t_list, name=None):
  return Tensor(torch.sqrt(sum((raw(t).detach() ** 2).sum() for t in t_list if t is not None)))


# ------------------------------------------------------------------------------------------------------------------
# contrib.framework: arg_scope
# ------------------------------------------------------------------------------------------------------------------
_ARG_STACK = [{}]


def _key(fn):
  return getattr(fn, '_key_op', None) or (fn.__module__, fn.__name__)


def add_arg_scope(func):
  @functools.wraps(func)
  def with_args(*args, **kwargs):
    defaults = _ARG_STACK[-1].get(_key(with_args))
    if defaults:
      merged = dict(defaults)
      merged.update(kwargs)
      kwargs = merged
    return func(*args, **kwargs)
  with_args._key_op = (func.__module__, func.__name__)
  return with_args


@contextlib.contextmanager
def arg_scope(list_ops_or_scope, **kwargs):
  if isinstance(list_ops_or_scope, dict):
    if kwargs:
      raise ValueError('When attempting to re-use a scope by suppling a dictionary, kwargs must be empty.')
    new = {k: dict(v) for k, v in list_ops_or_scope.items()}
  else:
    new = {k: dict(v) for k, v in _ARG_STACK[-1].items()}
    for op in list_ops_or_scope:
      if not hasattr(op, '_key_op'):
        raise ValueError('%s is not decorated with @add_arg_scope' % (op,))
      d = new.setdefault(_key(op), {})
      d.update(kwargs)
  _ARG_STACK.append(new)
  try:
    yield new
  finally:
    _ARG_STACK.pop()


def has_arg_scope(func):
  return hasattr(func, '_key_op')


@add_arg_scope
def model_variable(name, shape=None, dtype=float32, initializer=None, regularizer=None, trainable=True,
                   collections=None, caching_device=None, device=None, partitioner=None, custom_getter=None,
                   use_resource=None):
  collections = list(collections or []) + ['variables', 'model_variables']
  return core.get_variable(name, shape, dtype, initializer, regularizer, trainable, collections)


def get_unique_variable(var_op_name):
  """slim.get_unique_variable (contrib/framework/python/ops/variables.py): the one variable whose op name is exactly
  `var_op_name`; ValueError when there is none (image_generation.py:566-570 relies on both)."""
  for name, var in STATE.variables.items():
    if name == var_op_name:
      return var
  raise ValueError('Couldn\'t find variable %s' % var_op_name)


@add_arg_scope
def contrib_variable(name, shape=None, dtype=float32, initializer=None, regularizer=None, trainable=True,
                     collections=None, **unused):
  return core.get_variable(name, shape, dtype, initializer, regularizer, trainable, collections)


# ------------------------------------------------------------------------------------------------------------------
# contrib.layers
# ------------------------------------------------------------------------------------------------------------------
def _two(v):
  if isinstance(v, (list, tuple)):
    return [int(v[0]), int(v[1])]
  return [int(v), int(v)]


def get_variable_collections(variables_collections, name):
  if isinstance(variables_collections, dict):
    return variables_collections.get(name, None)
  return variables_collections


def collect_named_outputs(collections, alias, outputs):
  return outputs


def constant_value(value_or_tensor_or_var, dtype=None):
  v = value_or_tensor_or_var
  if isinstance(v, Tensor):
    return v.t.item() if v.t.dim() == 0 and not isinstance(v, Variable) else None
  return v


def smart_cond(pred, fn1, fn2, name=None):
  p = constant_value(pred)
  if p is None:
    p = bool(raw(pred).item())
  return fn1() if p else fn2()


def _apply_regularizer(regularizer, var):
  if regularizer is not None:
    r = regularizer(var)
    if r is not None:
      core.add_to_collection(core.GraphKeys.REGULARIZATION_LOSSES, r)


@add_arg_scope
def layers_convolution(inputs, num_outputs, kernel_size, stride=1, padding='SAME', data_format=None, rate=1,
                       activation_fn='relu', normalizer_fn=None, normalizer_params=None,
                       weights_initializer=None, weights_regularizer=None, biases_initializer=core.zeros_initializer(),
                       biases_regularizer=None, reuse=None, variables_collections=None, outputs_collections=None,
                       trainable=True, scope=None):
  """tf.contrib.layers.conv2d: variables `weights` [kh, kw, in, out] and (only without a normalizer_fn) `biases`;
  normalizer, then activation.  Default scope name 'Conv', uniquified like variable_scope(None, default_name)."""
  assert rate == 1 and data_format in (None, 'NHWC')
  with core.variable_scope(scope, 'Conv', [inputs], reuse=reuse) as sc:
    inputs = convert_to_tensor(inputs)
    kh, kw = _two(kernel_size)
    cin = int(inputs.shape[-1])
    # contrib's layers create every variable through _build_variable_getter -> slim's model_variable
    w = model_variable('weights', [kh, kw, cin, int(num_outputs)], inputs.dtype.base_dtype,
                       weights_initializer or core.glorot_uniform_initializer(), trainable=trainable,
                       collections=get_variable_collections(variables_collections, 'weights'))
    _apply_regularizer(weights_regularizer, w)
    sh, sw = _two(stride)
    out = nn_conv2d(inputs, w, [1, sh, sw, 1], padding)
    if normalizer_fn is None and biases_initializer is not None:
      b = model_variable('biases', [int(num_outputs)], inputs.dtype.base_dtype, biases_initializer,
                         trainable=trainable, collections=get_variable_collections(variables_collections, 'biases'))
      out = nn_bias_add(out, b)
    if normalizer_fn is not None:
      out = normalizer_fn(out, **(normalizer_params or {}))
    if activation_fn == 'relu':      # the contrib default, tf.nn.relu
      activation_fn = nn_relu
    if activation_fn is not None:
      out = activation_fn(out)
    return out


@add_arg_scope
def layers_fully_connected(inputs, num_outputs, activation_fn='relu', normalizer_fn=None, normalizer_params=None,
                           weights_initializer=None, weights_regularizer=None,
                           biases_initializer=core.zeros_initializer(), biases_regularizer=None, reuse=None,
                           variables_collections=None, outputs_collections=None, trainable=True, scope=None):
  """tf.contrib.layers.fully_connected: `weights` [in, out] (+ `biases` without a normalizer_fn); acts on the last
  axis.  Default scope name 'fully_connected'."""
  with core.variable_scope(scope, 'fully_connected', [inputs], reuse=reuse) as sc:
    inputs = convert_to_tensor(inputs)
    cin = int(inputs.shape[-1])
    w = model_variable('weights', [cin, int(num_outputs)], inputs.dtype.base_dtype,
                       weights_initializer or core.glorot_uniform_initializer(), trainable=trainable,
                       collections=get_variable_collections(variables_collections, 'weights'))
    _apply_regularizer(weights_regularizer, w)
    out = wrap(torch.matmul(raw(inputs), raw(w)), inputs)
    if normalizer_fn is None and biases_initializer is not None:
      b = model_variable('biases', [int(num_outputs)], inputs.dtype.base_dtype, biases_initializer,
                         trainable=trainable, collections=get_variable_collections(variables_collections, 'biases'))
      out = nn_bias_add(out, b)
    if normalizer_fn is not None:
      out = normalizer_fn(out, **(normalizer_params or {}))
    if activation_fn == 'relu':      # the contrib default, tf.nn.relu
      activation_fn = nn_relu
    if activation_fn is not None:
      out = activation_fn(out)
    return out


@add_arg_scope
def layers_layer_norm(inputs, center=True, scale=True, activation_fn=None, reuse=None, variables_collections=None,
                      outputs_collections=None, trainable=True, begin_norm_axis=1, begin_params_axis=-1, scope=None):
  """tf.contrib.layers.layer_norm as TF 1.8 ships it (contrib/layers/python/layers/layers.py), restated: `beta` (zeros)
  and `gamma` (ones) model variables of shape inputs.shape[begin_params_axis:] under variable_scope(scope, 'LayerNorm');
  moments over axes [begin_norm_axis, rank) kept as dims; tf.nn.batch_normalization with variance_epsilon 1e-12."""
  with core.variable_scope(scope, 'LayerNorm', [inputs], reuse=reuse):
    inputs = convert_to_tensor(inputs)
    rank = raw(inputs).dim()
    if begin_norm_axis < 0:
      begin_norm_axis += rank
    params_shape = [int(d) for d in inputs.shape[begin_params_axis:]]
    dt = inputs.dtype.base_dtype
    beta = model_variable('beta', params_shape, dt, core.zeros_initializer(), trainable=trainable) if center else None
    gamma = model_variable('gamma', params_shape, dt, core.ones_initializer(), trainable=trainable) if scale else None
    mean, variance = nn_moments(inputs, list(range(begin_norm_axis, rank)), keep_dims=True)
    out = nn_batch_normalization(inputs, mean, variance, beta, gamma, 1e-12)
    return activation_fn(out) if activation_fn is not None else out


@add_arg_scope
def layers_batch_norm(inputs, decay=0.999, center=True, scale=False, epsilon=0.001, activation_fn=None,
                      param_initializers=None, param_regularizers=None, updates_collections='update_ops',
                      is_training=True, reuse=None, variables_collections=None, outputs_collections=None, trainable=True,
                      batch_weights=None, fused=None, data_format='NHWC', zero_debias_moving_mean=False, scope=None,
                      renorm=False, renorm_clipping=None, renorm_decay=0.99, adjustment=None):
  """tf.contrib.layers.batch_norm on the route TF 1.8 takes with renorm=True (contrib's layers.py hands a call without
  batch_weights / zero-debias and with the default updates collection to the core tf.layers.BatchNormalization, whose
  non-fused call() is python/layers/normalization.py), restated.  Variables under variable_scope(scope, 'BatchNorm'):
  gamma, beta, moving_mean, moving_variance and, with renorm, renorm_mean, renorm_mean_weight (scalar), renorm_stddev,
  renorm_stddev_weight (scalar), all zero-initialised but gamma / moving_variance.  Training: batch moments; with renorm
  r = clip(sigma / mixed_sigma), d = clip((mu - mixed_mu) / mixed_sigma) against the pre-update averages 'as if they were
  initialised with this batch' (mixed_x = renorm_x + (1 - renorm_x_weight) * batch_x), stop-gradient, folded into scale /
  offset; renorm averages and their weights decay with renorm_decay, the moving mean / variance follow the de-biased
  renorm values with `decay`.  Inference: the moving statistics, r = 1, d = 0."""
  assert batch_weights is None and not zero_debias_moving_mean and adjustment is None and data_format == 'NHWC'
  assert not param_initializers and not param_regularizers and updates_collections == 'update_ops'
  with core.variable_scope(scope, 'BatchNorm', [inputs], reuse=reuse):
    x = convert_to_tensor(inputs)
    c, dt = int(x.shape[-1]), x.dtype.base_dtype

    def var(name, shape, init, train=False):
      return model_variable(name, shape, dt, init, trainable=train and trainable)

    gamma = var('gamma', [c], core.ones_initializer(), True) if scale else None
    beta = var('beta', [c], core.zeros_initializer(), True) if center else None
    moving_mean = var('moving_mean', [c], core.zeros_initializer())
    moving_variance = var('moving_variance', [c], core.ones_initializer())
    if renorm:
      unknown = set(renorm_clipping or {}) - {'rmax', 'rmin', 'dmax'}
      if unknown:
        raise ValueError('renorm_clipping contains keys not in rmax / rmin / dmax: %s' % sorted(unknown))
      r_mean, r_mean_w = var('renorm_mean', [c], core.zeros_initializer()), var('renorm_mean_weight', [], core.zeros_initializer())
      r_std, r_std_w = var('renorm_stddev', [c], core.zeros_initializer()), var('renorm_stddev_weight', [], core.zeros_initializer())
    training = constant_value(is_training)
    training = bool(raw(is_training).item()) if training is None else bool(training)
    sc_t = raw(gamma) if gamma is not None else None
    off_t = raw(beta) if beta is not None else None
    if training:
      mean, variance = nn_moments(x, [0, 1, 2][:raw(x).dim() - 1])
      new_mean, new_variance = raw(mean).detach(), raw(variance).detach()
      if renorm:
        clip = {k: raw(v) if isinstance(v, Tensor) else v for k, v in (renorm_clipping or {}).items()}
        with torch.no_grad():
          mu, sigma = raw(mean).detach(), torch.sqrt(raw(variance).detach() + epsilon)
          mixed_mu = r_mean.t + (1.0 - r_mean_w.t) * mu
          mixed_sigma = r_std.t + (1.0 - r_std_w.t) * sigma
          r, d = sigma / mixed_sigma, (mu - mixed_mu) / mixed_sigma
          if clip.get('rmin') is not None:
            r = torch.maximum(r, torch.as_tensor(clip['rmin'], dtype=r.dtype))
          if clip.get('rmax') is not None:
            r = torch.minimum(r, torch.as_tensor(clip['rmax'], dtype=r.dtype))
          if clip.get('dmax') is not None:
            dm = torch.as_tensor(clip['dmax'], dtype=d.dtype)
            d = torch.minimum(torch.maximum(d, -dm), dm)
          one = torch.ones((), dtype=mu.dtype)
          new_mean = raw(assign_moving_average(r_mean, mu, renorm_decay, False)) / raw(assign_moving_average(r_mean_w, one, renorm_decay, False))
          new_std = raw(assign_moving_average(r_std, sigma, renorm_decay, False)) / raw(assign_moving_average(r_std_w, one, renorm_decay, False))
          new_variance = new_std * new_std - epsilon
        off_t = d * (sc_t if sc_t is not None else 1.0) + (off_t if off_t is not None else 0.0)
        sc_t = r * (sc_t if sc_t is not None else 1.0)
      assign_moving_average(moving_mean, new_mean, decay, False)
      assign_moving_average(moving_variance, new_variance, decay, False)
    else:
      mean, variance = moving_mean, moving_variance
    out = nn_batch_normalization(x, mean, variance, None if off_t is None else wrap(off_t, x),
                                 None if sc_t is None else wrap(sc_t, x), epsilon)
    return activation_fn(out) if activation_fn is not None else out


def nn_relu(features, name=None):
  return wrap(torch.relu(raw(features)), features)


def l2_regularizer(scale, scope=None):
  def reg(weights):
    return Tensor(float(scale) * 0.5 * (raw(weights) ** 2).sum(), float32, 'l2_regularizer')
  return reg


def _unsupported(name):
  def fn(*a, **k):
    raise NotImplementedError('tf shim: %s is not implemented' % name)
  fn.__name__ = name.split('.')[-1]
  fn.__module__ = 'tensorflow.contrib.layers.python.layers.layers'
  return add_arg_scope(fn)


class _LayerVariableGetter(object):
  """contrib layers._build_variable_getter(rename) -> _model_variable_getter: every tf.get_variable made under the
  layer's scope -- the layer's own kernel / bias AND whatever its call() creates, e.g. the spectral-norm ``u`` of
  libs/sn.py:56 -- is renamed by its last path component and created through slim's model_variable, which does
  ``list(collections or []) + [GLOBAL_VARIABLES, MODEL_VARIABLES]`` (so the bare string libs/sn.py passes becomes a
  list of its characters, and the variable lands in MODEL_VARIABLES all the same)."""
  layer_getter = True

  def __init__(self, rename=None):
    self.rename = dict(rename or {})

  def __call__(self, name, shape, dtype, initializer, regularizer, trainable, collections):
    short = name.split('/')[-1]
    if short in self.rename:
      name = '/'.join(name.split('/')[:-1] + [self.rename[short]])
    return model_variable(name, shape=shape, dtype=dtype or float32, initializer=initializer, regularizer=regularizer,
                          trainable=trainable, collections=collections)


def _build_variable_getter(rename=None):
  return _LayerVariableGetter(rename)


# ---- tf.layers base classes (libs/sn.py subclasses them) ----------------------------------------------------------
_RENAME = {'kernel': 'weights', 'bias': 'biases'}      # what layers._build_variable_getter({...}) does in contrib


class _Layer(object):
  """tf.layers.Layer as contrib.layers uses it: `apply` = build once inside the captured variable scope, then call."""

  def __init__(self, trainable=True, name=None, dtype=None, activity_regularizer=None, _scope=None, _reuse=None,
               **unused):
    self.trainable, self.name, self.dtype = trainable, name, dtype or float32
    self._scope, self._reuse, self.built = _scope, _reuse, False

  def add_variable(self, name, shape, initializer=None, regularizer=None, trainable=True, dtype=None):
    v = core.get_variable(_RENAME.get(name, name), shape, dtype or self.dtype, initializer,
                          trainable=trainable and self.trainable)
    _apply_regularizer(regularizer, v)
    return v

  def apply(self, inputs):
    with core.variable_scope(self._scope, reuse=self._reuse):
      if not self.built:
        self.build(inputs.shape)
        self.built = True
      return self.call(inputs)

  __call__ = apply


class Convolution2D(_Layer):
  rank = 2

  def __init__(self, filters, kernel_size, strides=1, padding='valid', data_format='channels_last', dilation_rate=1,
               activation=None, use_bias=True, kernel_initializer=None, bias_initializer=None, kernel_regularizer=None,
               bias_regularizer=None, **kw):
    _Layer.__init__(self, **kw)
    assert data_format == 'channels_last' and _two(dilation_rate) == [1, 1]
    self.filters, self.kernel_size, self.strides = int(filters), _two(kernel_size), _two(strides)
    self.padding, self.data_format, self.activation, self.use_bias = padding, data_format, activation, bool(use_bias)
    self.kernel_initializer, self.bias_initializer = kernel_initializer, bias_initializer
    self.kernel_regularizer, self.bias_regularizer = kernel_regularizer, bias_regularizer
    self.kernel = self.bias = None

  def build(self, input_shape):
    cin = int(input_shape[-1])
    self.kernel = self.add_variable('kernel', self.kernel_size + [cin, self.filters], self.kernel_initializer,
                                    self.kernel_regularizer)
    if self.use_bias:
      self.bias = self.add_variable('bias', [self.filters], self.bias_initializer, self.bias_regularizer)

  def _convolution_op(self, inputs, kernel):
    return nn_conv2d(inputs, kernel, [1] + self.strides + [1], self.padding.upper())

  def call(self, inputs):
    out = self._convolution_op(inputs, self.kernel)
    if self.use_bias:
      out = nn_bias_add(out, self.bias)
    return self.activation(out) if self.activation is not None else out


class Dense(_Layer):
  def __init__(self, units, activation=None, use_bias=True, kernel_initializer=None, bias_initializer=None,
               kernel_regularizer=None, bias_regularizer=None, **kw):
    _Layer.__init__(self, **kw)
    self.units, self.activation, self.use_bias = int(units), activation, bool(use_bias)
    self.kernel_initializer, self.bias_initializer = kernel_initializer, bias_initializer
    self.kernel_regularizer, self.bias_regularizer = kernel_regularizer, bias_regularizer
    self.kernel = self.bias = None

  def build(self, input_shape):
    self.kernel = self.add_variable('kernel', [int(input_shape[-1]), self.units], self.kernel_initializer,
                                    self.kernel_regularizer)
    if self.use_bias:
      self.bias = self.add_variable('bias', [self.units], self.bias_initializer, self.bias_regularizer)

  def call(self, inputs):
    out = wrap(torch.matmul(raw(inputs), raw(self.kernel)), inputs)
    if self.use_bias:
      out = nn_bias_add(out, self.bias)
    return self.activation(out) if self.activation is not None else out


def _not_a_layer(name):
  class Unsupported(_Layer):
    def __init__(self, *a, **k):
      raise NotImplementedError('tf shim: %s' % name)
  return Unsupported


# ------------------------------------------------------------------------------------------------------------------
# module tree
# ------------------------------------------------------------------------------------------------------------------
class StubObject(object):
  """Anything the hot path never really uses (summaries, savers, queues ...): absorbs calls and attribute access."""

  def __init__(self, name):
    object.__setattr__(self, '_name', name)

  def __getattr__(self, k):
    if k.startswith('__') and k.endswith('__'):
      raise AttributeError(k)
    return StubObject(self._name + '.' + k)

  def __call__(self, *a, **k):
    return StubObject(self._name + '()')

  def __enter__(self):
    return self

  def __exit__(self, *a):
    return False

  def __iter__(self):
    return iter(())

  def __repr__(self):
    return '<stub %s>' % self._name


class StubModule(types.ModuleType):
  def __getattr__(self, k):
    if k.startswith('__') and k.endswith('__'):
      raise AttributeError(k)
    v = sys.modules.get(self.__name__ + '.' + k) or StubObject(self.__name__ + '.' + k)
    setattr(self, k, v)
    return v


def _module(name, **attrs):
  m = StubModule(name)
  m.__path__ = []      # a package: submodule imports go through the finder below
  for k, v in attrs.items():
    setattr(m, k, v)
  return m


def build_modules():
  logging = _module('tensorflow.logging', INFO=20, WARN=30, ERROR=40, DEBUG=10,
                    info=lambda *a, **k: None, warning=lambda *a, **k: None, warn=lambda *a, **k: None,
                    error=lambda *a, **k: None, log_every_n=lambda *a, **k: None, log=lambda *a, **k: None,
                    set_verbosity=lambda *a, **k: None)
  nn = _module('tensorflow.nn', relu=nn_relu, leaky_relu=nn_leaky_relu, tanh=tanh, sigmoid=sigmoid,
               softmax=nn_softmax, bias_add=nn_bias_add, avg_pool=nn_avg_pool, conv2d=nn_conv2d, moments=nn_moments,
               batch_normalization=nn_batch_normalization, l2_normalize=nn_l2_normalize,
               sigmoid_cross_entropy_with_logits=nn_sigmoid_xent)
  image = _module('tensorflow.image', resize_nearest_neighbor=image_resize_nearest,
                  resize_bilinear=image_resize_bilinear, ResizeMethod=ResizeMethod,
                  convert_image_dtype=image_convert_image_dtype, pad_to_bounding_box=image_pad_to_bounding_box,
                  crop_to_bounding_box=image_crop_to_bounding_box, resize_images=image_resize_images,
                  adjust_saturation=image_adjust_saturation, random_brightness=image_random_brightness,
                  random_saturation=image_random_saturation, random_hue=_unsupported('image.random_hue'),
                  random_contrast=_unsupported('image.random_contrast'))
  control_flow_ops = _module('tensorflow.python.ops.control_flow_ops', switch=cf_switch, merge=cf_merge)
  losses = _module('tensorflow.losses', compute_weighted_loss=compute_weighted_loss,
                   absolute_difference=absolute_difference, sigmoid_cross_entropy=sigmoid_cross_entropy,
                   cosine_distance=cosine_distance, get_losses=get_losses, Reduction=Reduction,
                   get_regularization_losses=lambda scope=None: core.get_collection(
                     core.GraphKeys.REGULARIZATION_LOSSES, scope))
  train = _module('tensorflow.train', get_global_step=get_global_step,
                  get_or_create_global_step=get_or_create_global_step, piecewise_constant=piecewise_constant,
                  latest_checkpoint=lambda checkpoint_dir, latest_filename=None: None,
                  Optimizer=Optimizer, GradientDescentOptimizer=Optimizer, AdamOptimizer=AdamOptimizer)
  app = _module('tensorflow.app', flags=core.flags)

  conv2d = layers_convolution
  fully_connected = layers_fully_connected
  layers_impl = _module(
    'tensorflow.contrib.layers.python.layers.layers', conv2d=conv2d, convolution=conv2d, convolution2d=conv2d,
    fully_connected=fully_connected, conv2d_transpose=_unsupported('layers.conv2d_transpose'),
    batch_norm=layers_batch_norm, layer_norm=layers_layer_norm,
    instance_norm=_unsupported('layers.instance_norm'), l2_regularizer=l2_regularizer,
    xavier_initializer=core.glorot_uniform_initializer, utils=None,
    _build_variable_getter=_build_variable_getter, _add_variable_to_collections=lambda *a, **k: None,
    core_layers=_module('tensorflow.python.layers.core', Dense=Dense),
    six=_module('six', integer_types=(int, np.integer)), nn=None)
  utils = _module('tensorflow.contrib.layers.python.layers.utils', get_variable_collections=get_variable_collections,
                  collect_named_outputs=collect_named_outputs, smart_cond=smart_cond, constant_value=constant_value,
                  two_element_tuple=lambda v: tuple(_two(v)))
  initializers = _module('tensorflow.contrib.layers.python.layers.initializers',
                         xavier_initializer=core.glorot_uniform_initializer)
  layers_impl.utils = utils
  py_layers = _module('tensorflow.contrib.layers.python.layers', layers=layers_impl, utils=utils,
                      initializers=initializers, batch_norm=layers_impl.batch_norm, convolution=conv2d,
                      fully_connected=fully_connected)
  py = _module('tensorflow.contrib.layers.python', layers=py_layers)
  contrib_layers = _module(
    'tensorflow.contrib.layers', conv2d=conv2d, convolution=conv2d, convolution2d=conv2d,
    fully_connected=fully_connected, conv2d_transpose=layers_impl.conv2d_transpose, batch_norm=layers_impl.batch_norm,
    layer_norm=layers_impl.layer_norm, instance_norm=layers_impl.instance_norm, l2_regularizer=l2_regularizer,
    xavier_initializer=core.glorot_uniform_initializer, python=py)

  fw_variables = _module('tensorflow.contrib.framework.python.ops.variables', model_variable=model_variable,
                         variable=contrib_variable, get_or_create_global_step=get_or_create_global_step)
  fw_ops = _module('tensorflow.contrib.framework.python.ops', add_arg_scope=add_arg_scope, arg_scope=arg_scope,
                   variables=fw_variables, has_arg_scope=has_arg_scope)
  fw_py = _module('tensorflow.contrib.framework.python', ops=fw_ops)
  framework = _module('tensorflow.contrib.framework', arg_scope=arg_scope, add_arg_scope=add_arg_scope,
                      python=fw_py, model_variable=model_variable, get_or_create_global_step=get_or_create_global_step,
                      get_variables=lambda scope=None, suffix=None, collection='variables': core.get_collection(
                        collection, scope), get_model_variables=lambda scope=None, suffix=None: core.get_collection(
                        'model_variables', scope))
  slim = _module('tensorflow.contrib.slim', arg_scope=arg_scope, add_arg_scope=add_arg_scope, conv2d=conv2d,
                 fully_connected=fully_connected, model_variable=model_variable,
                 get_or_create_global_step=get_or_create_global_step, l2_regularizer=l2_regularizer,
                 variable=contrib_variable, get_variables=framework.get_variables, get_model_variables=framework.get_model_variables,
                 get_unique_variable=get_unique_variable)
  contrib = _module('tensorflow.contrib', layers=contrib_layers, framework=framework, slim=slim)

  py_fw_ops = _module('tensorflow.python.framework.ops', convert_to_tensor=convert_to_tensor,
                      add_to_collections=core.add_to_collections, add_to_collection=core.add_to_collection,
                      control_dependencies=control_dependencies, device=_null_context, colocate_with=_null_context,
                      name_scope=core.name_scope, Tensor=Tensor, GraphKeys=core.GraphKeys)
  array_ops = _module('tensorflow.python.ops.array_ops', constant=constant, identity=identity, reshape=reshape,
                      shape=shape, stop_gradient=stop_gradient, ones_like=ones_like, zeros_like=zeros_like,
                      expand_dims=expand_dims, squeeze=squeeze, concat=concat, transpose=transpose)
  convolutional = _module('tensorflow.python.layers.convolutional', Convolution2D=Convolution2D, Conv2D=Convolution2D,
                          Convolution1D=_not_a_layer('Convolution1D'), Convolution3D=_not_a_layer('Convolution3D'))
  gen_math_ops = _module('tensorflow.python.ops.gen_math_ops', mat_mul=matmul)
  moving_averages = _module('tensorflow.python.training.moving_averages',
                            assign_moving_average=assign_moving_average)
  context = _module('tensorflow.python.eager.context', executing_eagerly=lambda: False, in_eager_mode=lambda: False)

  tf = _module(
    'tensorflow', __version__='1.8.0-shim',
    float16=float16, float32=float32, float64=float64, int32=int32, int64=int64, bool=bool_, uint8=core.uint8,
    DType=core.DType, reverse=reverse, random_crop=random_crop, matrix_transpose=matrix_transpose,
    Tensor=Tensor, Variable=Variable, TensorShape=TensorShape, Dimension=Dimension,
    constant=constant, convert_to_tensor=convert_to_tensor, cast=cast, to_float=to_float, identity=identity,
    stop_gradient=stop_gradient, reshape=reshape, expand_dims=expand_dims, squeeze=squeeze, concat=concat, stack=stack,
    transpose=transpose, tile=tile, pad=pad, shape=shape, zeros_like=zeros_like, ones_like=ones_like, zeros=zeros,
    ones=ones, sqrt=sqrt, rsqrt=rsqrt, square=square, exp=exp, log=log, negative=negative, tanh=tanh, sigmoid=sigmoid,
    abs=abs_, add=add, subtract=subtract, multiply=multiply, divide=divide, div=divide, minimum=minimum,
    maximum=maximum, pow=pow_, add_n=add_n, clip_by_value=clip_by_value, where=where, equal=equal,
    not_equal=not_equal, greater=greater, less=less, greater_equal=greater_equal, less_equal=less_equal,
    mod=floormod, floormod=floormod, global_norm=global_norm, IndexedSlices=type('IndexedSlices', (), {}),
    reduce_mean=reduce_mean, reduce_sum=reduce_sum, reduce_max=reduce_max, reduce_min=reduce_min, matmul=matmul,
    tensordot=tensordot, random_normal=random_normal, random_uniform=random_uniform,
    control_dependencies=control_dependencies, device=_null_context, assign=assign, assign_add=assign_add,
    assign_sub=assign_sub, group=group, no_op=no_op, cond=cond, while_loop=while_loop, gradients=gradients,
    placeholder=placeholder, placeholder_with_default=placeholder_with_default,
    variable_scope=core.variable_scope, name_scope=core.name_scope, get_variable=core.get_variable,
    get_variable_scope=core.get_variable_scope, AUTO_REUSE=core.AUTO_REUSE, GraphKeys=core.GraphKeys,
    add_to_collection=core.add_to_collection, add_to_collections=core.add_to_collections,
    get_collection=core.get_collection, get_collection_ref=core.get_collection_ref,
    zeros_initializer=core.zeros_initializer, ones_initializer=core.ones_initializer,
    constant_initializer=core.constant_initializer, random_normal_initializer=core.random_normal_initializer,
    truncated_normal_initializer=core.truncated_normal_initializer,
    glorot_uniform_initializer=core.glorot_uniform_initializer,
    executing_eagerly=lambda: False, flags=core.flags, app=app, logging=logging, nn=nn, image=image, losses=losses,
    train=train, contrib=contrib,
    global_variables=lambda scope=None: core.get_collection('variables', scope),
    trainable_variables=lambda scope=None: core.get_collection('trainable_variables', scope),
    model_variables=lambda scope=None: core.get_collection('model_variables', scope))

  mods = {m.__name__: m for m in (
    tf, logging, nn, image, losses, train, app, contrib, contrib_layers, py, py_layers, layers_impl, utils,
    initializers, framework, fw_py, fw_ops, fw_variables, slim, py_fw_ops, array_ops, context, convolutional,
    gen_math_ops, moving_averages, control_flow_ops)}
  return mods


class _StubFinder(object):
  """Resolves every other `tensorflow.*` (and explicitly stubbed) import to an absorbing stub module."""

  def __init__(self, prefixes):
    self.prefixes = tuple(prefixes)

  def find_spec(self, name, path=None, target=None):
    import importlib.machinery
    if name.startswith(self.prefixes) or name in self.prefixes:
      return importlib.machinery.ModuleSpec(name, self, is_package=True)
    return None

  def create_module(self, spec):
    m = StubModule(spec.name)
    m.__path__ = []
    return m

  def exec_module(self, module):
    pass


def install(extra_stub_prefixes=()):
  """Puts the shim's `tensorflow` into sys.modules.  Refuses to shadow a real TensorFlow."""
  if 'tensorflow' in sys.modules and not isinstance(sys.modules['tensorflow'], StubModule):
    raise RuntimeError('a real tensorflow is already imported')
  mods = build_modules()
  sys.modules.update(mods)
  sys.meta_path.insert(0, _StubFinder(('tensorflow.',) + tuple(extra_stub_prefixes)))
  return mods['tensorflow']
