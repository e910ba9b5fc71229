"""float64 NumPy restatement of every primitive on the TwinGAN hot path (oracle, test-only).

Parity status: these are TensorFlow-op semantics restated (not pinned by a TensorFlow run) -- see
``oracle/__init__.py``.  All tensors are NHWC, weights HWIO
``[kh, kw, Cin, Cout]`` exactly as the reference's TF variables.  Each function cites the
reference lines (relative to /root/reference) it restates.
"""
import numpy as np

F64 = np.float64


def get_num_channels(stage, max_num_channels=256):
  """nets/pggan_utils.py:369-372 (python-2 integer division)."""
  return min(1024 // (2 ** stage), max_num_channels)


# --------------------------------------------------------------------------------------------
# conv / fc  (tf.contrib.layers.conv2d -> tf.nn.conv2d, called at nets/pggan_utils.py:316-320)
# --------------------------------------------------------------------------------------------
def same_pads(k):
  """TF 'SAME', stride 1: total pad k-1, low side gets floor((k-1)/2)."""
  lo = (k - 1) // 2
  return lo, (k - 1) - lo


def conv2d(x, w, padding='SAME'):
  """Stride-1 cross-correlation, NHWC x HWIO.  nets/pggan_utils.py:95-97 (stride 1, SAME, k=3)."""
  x = np.asarray(x, F64)
  w = np.asarray(w, F64)
  n, h, ww, cin = x.shape
  kh, kw, cin2, cout = w.shape
  assert cin == cin2
  if padding == 'SAME':
    (pt, pb), (pl, pr) = same_pads(kh), same_pads(kw)
  else:
    pt = pb = pl = pr = 0
  xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
  ho = xp.shape[1] - kh + 1
  wo = xp.shape[2] - kw + 1
  y = np.zeros((n, ho, wo, cout), F64)
  for i in range(kh):
    for j in range(kw):
      y += np.einsum('nhwc,co->nhwo', xp[:, i:i + ho, j:j + wo, :], w[i, j])
  return y


def conv2d_bwd_data(gy, w, in_hw, padding='SAME'):
  """d conv2d / d x  (TF Conv2DBackpropInput)."""
  gy = np.asarray(gy, F64)
  w = np.asarray(w, F64)
  kh, kw, cin, cout = w.shape
  n, ho, wo, _ = gy.shape
  h, ww = in_hw
  if padding == 'SAME':
    (pt, pb), (pl, pr) = same_pads(kh), same_pads(kw)
  else:
    pt = pb = pl = pr = 0
  gxp = np.zeros((n, h + pt + pb, ww + pl + pr, cin), F64)
  for i in range(kh):
    for j in range(kw):
      gxp[:, i:i + ho, j:j + wo, :] += np.einsum('nhwo,co->nhwc', gy, w[i, j])
  return gxp[:, pt:pt + h, pl:pl + ww, :]


def conv2d_bwd_weight(x, gy, ksize, padding='SAME'):
  """d conv2d / d w  (TF Conv2DBackpropFilter)."""
  x = np.asarray(x, F64)
  gy = np.asarray(gy, F64)
  kh, kw = ksize
  n, ho, wo, cout = gy.shape
  cin = x.shape[3]
  if padding == 'SAME':
    (pt, pb), (pl, pr) = same_pads(kh), same_pads(kw)
  else:
    pt = pb = pl = pr = 0
  xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
  gw = np.zeros((kh, kw, cin, cout), F64)
  for i in range(kh):
    for j in range(kw):
      gw[i, j] = np.einsum('nhwc,nhwo->co', xp[:, i:i + ho, j:j + wo, :], gy)
  return gw


def _tap_views(xp, kh, kw, ho, wo):
  for i in range(kh):
    for j in range(kw):
      yield i, j, xp[:, i:i + ho, j:j + wo, :]


def conv2d_gemm(x, w, padding='SAME'):
  """conv2d() with every tap as one float64 BLAS product ([pixels, Cin] @ [Cin, Cout]) instead of an einsum loop:
  the same sums, fast enough for the full-size layer shapes of tests/test_gpu_bench_shapes.py.  Held to conv2d() by
  tests/test_oracle.py."""
  x = np.asarray(x, F64)
  w = np.asarray(w, F64)
  n, h, ww, cin = x.shape
  kh, kw, _, cout = w.shape
  (pt, pb), (pl, pr) = (same_pads(kh), same_pads(kw)) if padding == 'SAME' else ((0, 0), (0, 0))
  xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
  ho, wo = xp.shape[1] - kh + 1, xp.shape[2] - kw + 1
  y = np.zeros((n * ho * wo, cout), F64)
  for i, j, v in _tap_views(xp, kh, kw, ho, wo):
    y += np.ascontiguousarray(v).reshape(-1, cin) @ w[i, j]
  return y.reshape(n, ho, wo, cout)


def conv2d_bwd_data_gemm(gy, w, in_hw, padding='SAME'):
  """conv2d_bwd_data() as the correlation of the padded gradient with the 180-degree rotated, transposed kernel."""
  gy = np.asarray(gy, F64)
  w = np.asarray(w, F64)
  kh, kw, cin, cout = w.shape
  n, ho, wo, _ = gy.shape
  h, ww = in_hw
  (pt, pb), (pl, pr) = (same_pads(kh), same_pads(kw)) if padding == 'SAME' else ((0, 0), (0, 0))
  # gx[y, x] = sum_{i,j} gy[y + pt - i, x + pl - j] w[i, j]^T: pad gy so that every index is in range
  gp = np.pad(gy, ((0, 0), (kh - 1 - pt, h - ho + pt), (kw - 1 - pl, ww - wo + pl), (0, 0)))
  gx = np.zeros((n * h * ww, cin), F64)
  for i in range(kh):
    for j in range(kw):
      v = gp[:, kh - 1 - i:kh - 1 - i + h, kw - 1 - j:kw - 1 - j + ww, :]
      gx += np.ascontiguousarray(v).reshape(-1, cout) @ w[i, j].T
  return gx.reshape(n, h, ww, cin)


def conv2d_bwd_weight_gemm(x, gy, ksize, padding='SAME', per_image=False):
  """conv2d_bwd_weight() with one BLAS product per tap.  ``per_image``: [N, kh, kw, Cin, Cout], the per-image terms
  whose sum over N is the filter gradient (any sub-batch / pair of segments is then a sum of slices)."""
  x = np.asarray(x, F64)
  gy = np.asarray(gy, F64)
  kh, kw = ksize
  n, ho, wo, cout = gy.shape
  cin = x.shape[3]
  (pt, pb), (pl, pr) = (same_pads(kh), same_pads(kw)) if padding == 'SAME' else ((0, 0), (0, 0))
  xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
  if per_image:
    gw = np.zeros((n, kh, kw, cin, cout), F64)
    g2 = gy.reshape(n, ho * wo, cout)
    for i, j, v in _tap_views(xp, kh, kw, ho, wo):
      gw[:, i, j] = np.matmul(np.ascontiguousarray(v).reshape(n, ho * wo, cin).transpose(0, 2, 1), g2)
    return gw
  gw = np.zeros((kh, kw, cin, cout), F64)
  g2 = gy.reshape(-1, cout)
  for i, j, v in _tap_views(xp, kh, kw, ho, wo):
    gw[i, j] = np.ascontiguousarray(v).reshape(-1, cin).T @ g2
  return gw


def fully_connected(x, w, b=None):
  """layers.fully_connected, nets/pggan_utils.py:323-327; weights [in, out]."""
  y = np.asarray(x, F64) @ np.asarray(w, F64)
  if b is not None:
    y = y + np.asarray(b, F64)
  return y


# --------------------------------------------------------------------------------------------
# pointwise / normalisation
# --------------------------------------------------------------------------------------------
def leaky_relu(x, alpha=0.2):
  """util_misc.py:68-86: tf.maximum(alpha * x, x)."""
  x = np.asarray(x, F64)
  return np.maximum(alpha * x, x)


def pixel_norm(x, eps=1e-6):
  """nets/pggan_utils.py:330-331: x / sqrt(mean_C(x^2) + eps)."""
  x = np.asarray(x, F64)
  return x / np.sqrt(np.mean(np.square(x), axis=3, keepdims=True) + eps)


def instance_norm(x, gamma, beta, eps=1e-6):
  """libs/instance_norm.py:131-135: tf.nn.moments over (H,W) (biased variance), then
  tf.nn.batch_normalization: inv = rsqrt(var+eps)*gamma ; y = x*inv + (beta - mean*inv)."""
  x = np.asarray(x, F64)
  mean = x.mean(axis=(1, 2), keepdims=True)
  var = np.mean(np.square(x - mean), axis=(1, 2), keepdims=True)
  inv = 1.0 / np.sqrt(var + eps) * np.asarray(gamma, F64)
  return x * inv + (np.asarray(beta, F64) - mean * inv)


def batch_norm_train(x, gamma, beta, eps=1e-3):
  """libs/batch_norm.py:430,464-470 (training mode, no renorm): batch moments over (N,H,W),
  epsilon = max(1e-3, 1.001e-5) (libs/batch_norm.py:48,464-468)."""
  x = np.asarray(x, F64)
  mean = x.mean(axis=(0, 1, 2), keepdims=True)
  var = np.mean(np.square(x - mean), axis=(0, 1, 2), keepdims=True)
  inv = 1.0 / np.sqrt(var + eps) * np.asarray(gamma, F64)
  return x * inv + (np.asarray(beta, F64) - mean * inv)


def bias_add(x, b):
  return np.asarray(x, F64) + np.asarray(b, F64)


def upsample2x(x):
  """nets/pggan_utils.py:349-350: tf.image.resize_nearest_neighbor to (2h, 2w): out[i,j]=in[i//2,j//2]."""
  x = np.asarray(x, F64)
  return np.repeat(np.repeat(x, 2, axis=1), 2, axis=2)


def avg_pool2(x):
  """nets/pggan.py:274,306,436,468: tf.nn.avg_pool 2x2 stride 2 VALID."""
  x = np.asarray(x, F64)
  n, h, w, c = x.shape
  return x.reshape(n, h // 2, 2, w // 2, 2, c).mean(axis=(2, 4))


def lerp(new, old, alpha):
  """nets/pggan.py:205,314,475: new * alpha + (1 - alpha) * old."""
  return np.asarray(new, F64) * alpha + (1.0 - alpha) * np.asarray(old, F64)


def minibatch_state_concat(x, eps=1e-8):
  """nets/pggan_utils.py:353-366: sqrt(biased var over batch + eps) -> mean over everything ->
  one scalar tiled to [N,4,4,1] and concatenated on C (4x4 is hard-coded in the reference)."""
  x = np.asarray(x, F64)
  mean = x.mean(axis=0, keepdims=True)
  std = np.sqrt(np.mean(np.square(x - mean), axis=0, keepdims=True) + eps)
  val = std.mean()
  tile = np.full((x.shape[0], 4, 4, 1), val, F64)
  return np.concatenate([x, tile], axis=3)


# --------------------------------------------------------------------------------------------
# losses (twingan.py:451-505, image_generation.py:318-439)
# --------------------------------------------------------------------------------------------
def absolute_difference(labels, predictions, weight=1.0):
  """tf.losses.absolute_difference with scalar weight: mean(|p - l|) * w."""
  return np.mean(np.abs(np.asarray(predictions, F64) - np.asarray(labels, F64))) * weight


def wgan_d_loss(fake_pred, real_pred):
  """image_generation.py:350: mean D(fake) - mean D(real)."""
  return np.mean(fake_pred) - np.mean(real_pred)


def wgan_g_loss(fake_pred):
  """image_generation.py:333: -mean D(fake)."""
  return -np.mean(fake_pred)


def sigmoid_cross_entropy(labels, logits, weight=1.0):
  """tf.losses.sigmoid_cross_entropy (mean over elements) * weight, in TF's stable form
  max(x,0) - x*z + log(1 + exp(-|x|))  (image_generation.py:340-344,383-394)."""
  x, z = np.asarray(logits, F64), np.asarray(labels, F64)
  return np.mean(np.maximum(x, 0.0) - x * z + np.log1p(np.exp(-np.abs(x)))) * weight


def hinge_d_loss(fake_pred, real_pred):
  """image_generation.py:373-375: mean relu(1 + D(fake)) + mean relu(1 - D(real))."""
  f, r = np.asarray(fake_pred, F64), np.asarray(real_pred, F64)
  return np.mean(np.maximum(1.0 + f, 0.0)) + np.mean(np.maximum(1.0 - r, 0.0))


def drift_loss(real_pred, weight):
  """image_generation.py:360-367: w * mean(D(real)^2)."""
  return weight * np.mean(np.square(np.asarray(real_pred, F64)))


def dragan_perturbed_batch(minibatch, noise):
  """image_generation.py:441-449 (get_perturbed_batch): x + 0.5 * VARIANCE(all elements of x) * U(-1,1) -- the
  reference names it std but takes tf.nn.moments(...)[1], the variance.  ``noise`` = the U(-1,1) draw."""
  x = np.asarray(minibatch, F64)
  return x + 0.5 * np.var(x) * np.asarray(noise, F64)


def gradient_penalty(interp_grad, lam=10.0):
  """image_generation.py:431-436: slopes = sqrt(sum_{h,w,c} g^2) (no eps); mean((slopes-1)^2)*lambda."""
  g = np.asarray(interp_grad, F64)
  slopes = np.sqrt(np.sum(np.square(g), axis=(1, 2, 3)))
  return np.mean(np.square(slopes - 1.0)) * lam


# --------------------------------------------------------------------------------------------
# optimiser (model/model_inheritor.py:537-542 -> tf.train.AdamOptimizer)
# --------------------------------------------------------------------------------------------
def adam_step(theta, g, m, v, t, lr=1e-4, beta1=0.5, beta2=0.99, eps=1e-8):
  """TF-1.x Adam: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMA; theta -= lr_t*m/(sqrt(v)+eps).
  ``t`` is the 1-based count of applies (shared by G and D: image_generation.py:554-561)."""
  theta, g, m, v = (np.asarray(a, F64) for a in (theta, g, m, v))
  lr_t = lr * np.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
  m = beta1 * m + (1.0 - beta1) * g
  v = beta2 * v + (1.0 - beta2) * g * g
  theta = theta - lr_t * m / (np.sqrt(v) + eps)
  return theta, m, v


# ------------------------------------------------------------------------------------------------ input preprocessing
def resize_bilinear_tf1(x, out_h, out_w):
  """tf.image.resize_images(..., BILINEAR), align_corners=False, TF-1.x (no half-pixel centres): in = out * in_size /
  out_size; top-left = floor, bottom-right = min(+1, size - 1).  x: [h, w, c] float64."""
  h, w = x.shape[:2]
  # core/kernels/resize_bilinear_op.cc compute_interpolation_weights: scale, `in = i * scale` and `lerp = in - lower` are
  # float32 (it shows at ~1e-5 on the weights when in / out is not dyadic); the blend itself is evaluated in float64 here
  fy = np.arange(out_h, dtype=np.float32) * (np.float32(h) / np.float32(out_h))
  fx = np.arange(out_w, dtype=np.float32) * (np.float32(w) / np.float32(out_w))
  assert fy.dtype == np.float32 and fx.dtype == np.float32
  top, left = np.floor(fy).astype(int), np.floor(fx).astype(int)
  bot, right = np.minimum(top + 1, h - 1), np.minimum(left + 1, w - 1)
  ly = (fy - top.astype(np.float32)).astype(np.float64)[:, None, None]
  lx = (fx - left.astype(np.float32)).astype(np.float64)[None, :, None]
  t = x[top][:, left] + (x[top][:, right] - x[top][:, left]) * lx
  b = x[bot][:, left] + (x[bot][:, right] - x[bot][:, left]) * lx
  return t + (b - t) * ly


def rgb_to_hsv(rgb):
  """tf.image.rgb_to_hsv (core/kernels/colorspace_op.h), element-wise on [..., 3]."""
  r, g, b = rgb[..., 0], rgb[..., 1], rgb[..., 2]
  v = rgb.max(axis=-1)
  rng = v - rgb.min(axis=-1)
  s = np.where(v > 0, rng / np.where(v > 0, v, 1), 0.0)
  norm = 1.0 / (6.0 * np.where(rng > 0, rng, 1))
  h = np.where(r == v, norm * (g - b), np.where(g == v, norm * (b - r) + 2.0 / 6.0, norm * (r - g) + 4.0 / 6.0))
  h = np.where(rng > 0, h, 0.0)
  h = np.where(h < 0, h + 1.0, h)
  return np.stack([h, s, v], axis=-1)


def hsv_to_rgb(hsv):
  """tf.image.hsv_to_rgb (colorspace_op.h)."""
  h, s, v = hsv[..., 0], hsv[..., 1], hsv[..., 2]
  c = s * v
  m = v - c
  dh = h * 6.0
  cat = np.floor(dh).astype(int)
  fm = dh - 2.0 * np.floor(dh / 2.0)
  x = c * (1 - np.abs(fm - 1))
  z = np.zeros_like(c)
  table = [(c, x, z), (x, c, z), (z, c, x), (z, x, c), (x, z, c), (c, z, x)]
  out = np.zeros(hsv.shape)
  for k, (rr, gg, bb) in enumerate(table):
    sel = (cat == k) | ((k == 5) & (cat >= 6)) | ((k == 0) & (cat < 0))
    out[..., 0] = np.where(sel, rr, out[..., 0])
    out[..., 1] = np.where(sel, gg, out[..., 1])
    out[..., 2] = np.where(sel, bb, out[..., 2])
  return out + m[..., None]


def adjust_saturation(rgb, factor):
  """tf.image.adjust_saturation: HSV round trip with s = clip(s * factor, 0, 1)."""
  hsv = rgb_to_hsv(rgb)
  hsv[..., 1] = np.clip(hsv[..., 1] * factor, 0.0, 1.0)
  return hsv_to_rgb(hsv)


RGB_TO_YIQ = np.array([[0.299, 0.587, 0.114], [0.596, -0.274, -0.322], [0.211, -0.523, 0.312]], np.float32)      # preprocessing_util.py:154-160
RANDOM_CROP_RATIO = 0.8      # danbooru_preprocessing.py:33


def preprocess_image(img_u8, hw, resize_mode='PAD', is_training=True, flip=False, saturation_first=False, delta=0.0,
                     factor=1.0, crop=None, random_cropping_ratio=RANDOM_CROP_RATIO, color_space='rgb', mode_offset=None,
                     initial_crop_hw=None):
  """preprocessing/danbooru_preprocessing.py:115-230 as the TwinGAN trainer reaches it (model_inheritor.py:403-457:
  padding 0, no mean subtraction, fast_mode -- `fast_mode` is not a flag and the partial never sets it, so the hue /
  contrast orderings are unreachable from the trainer): convert_image_dtype to [0, 1]; resize_image
  (preprocessing_util.py:97-146) -- PAD: zero-pad about the centre to max(h, w); CROP: centre-crop to min(h, w); RESHAPE:
  as is -- then bilinear to [hw, hw]; when training: random_flip_left_right (flip if the draw < 0.5), distort_color in
  fast mode (ordering 0: brightness then saturation; orderings 1-3: saturation then brightness), clip to [0, 1].
  ``crop`` = (oy, ox, ch, cw): --do_random_cropping when training (docs/training.md:22-23 trains with it;
  danbooru_preprocessing.py:187-201, preprocessing_util.random_crop_image :312-331): the first resize goes to
  int(hw / random_cropping_ratio), tf.random_crop cuts the [ch, cw] rectangle at (oy, ox) out of it (ch, cw =
  int32(size * U[ratio, 1)), offsets uniform), a second bilinear resize brings that to [hw, hw].  ``color_space``:
  'gray' skips the colour distortion (:208-212); 'yiq' applies rgb_to_yiq, 'bgr' reverses the channels, after everything
  else (:221-225).  The random draws are arguments."""
  x = (img_u8.astype(np.float32) * np.float32(1.0 / 255.0)).astype(np.float64)
  h, w = x.shape[:2]
  if resize_mode == 'PAD' and h != w:
    size = max(h, w)
    oh, ow = (size - h) // 2, (size - w) // 2
    sq = np.zeros((size, size, x.shape[2]))
    sq[oh:oh + h, ow:ow + w] = x
    x = sq
  elif resize_mode == 'CROP' and h != w:
    size = min(h, w)
    oh, ow = (h - size) // 2, (w - size) // 2
    x = x[oh:oh + size, ow:ow + size]
  elif resize_mode == 'RANDOM_CROP':
    # preprocessing_util._random_crop_to_hw (:84-95): an image smaller than the target is first resized to it (the crop
    # is then the whole image); otherwise tf.random_crop cuts [new_hw, new_hw] at ``mode_offset`` = (oy, ox) and NO
    # resize follows (:144-146).  new_hw is the first resize's target: int(hw / ratio) with random cropping on.
    new_hw = int(hw / random_cropping_ratio) if (is_training and crop is not None) else hw
    if new_hw > min(h, w):
      x = resize_bilinear_tf1(x, new_hw, new_hw)
      oy = ox = 0
    else:
      oy, ox = (int(v) for v in mode_offset)
    assert oy + new_hw <= x.shape[0] and ox + new_hw <= x.shape[1]
    x = x[oy:oy + new_hw, ox:ox + new_hw]
  elif resize_mode == 'RANDOM_CROP_AND_RESHAPE':
    # :128-131: _random_crop_to_hw to --random_crop_and_reshape_initial_crop_hw (a smaller image is resized up to it
    # first), THEN the bilinear resize every mode but RANDOM_CROP ends with (:144-146)
    c = int(initial_crop_hw)
    assert c > 0
    if c > min(h, w):
      x = resize_bilinear_tf1(x, c, c)
      oy = ox = 0
    else:
      oy, ox = (int(v) for v in mode_offset)
    x = x[oy:oy + c, ox:ox + c]
  elif resize_mode == 'NONE':      # :137-139: the image as it is -- it must already have the size the networks take
    assert (h, w) == (hw, hw) and crop is None
  elif resize_mode not in ('PAD', 'CROP', 'RESHAPE'):
    raise ValueError(resize_mode)
  if is_training and crop is not None:
    mid = int(hw / random_cropping_ratio)
    x = resize_bilinear_tf1(x, mid, mid)
    oy, ox, ch, cw = (int(v) for v in crop)
    assert 0 <= oy and oy + ch <= mid and 0 <= ox and ox + cw <= mid
    x = resize_bilinear_tf1(x[oy:oy + ch, ox:ox + cw], hw, hw)
  else:
    x = resize_bilinear_tf1(x, hw, hw)
  if is_training:
    if flip:
      x = x[:, ::-1]
    if color_space != 'gray':
      if saturation_first:
        x = adjust_saturation(x, factor) + delta
      else:
        x = adjust_saturation(x + delta, factor)
      x = np.clip(x, 0.0, 1.0)
  if color_space == 'yiq':
    x = x @ RGB_TO_YIQ.astype(np.float64).T
  elif color_space == 'bgr':
    x = x[..., ::-1]
  elif color_space not in ('rgb', 'gray'):
    raise ValueError(color_space)
  return x


# ------------------------------------------------------------------------------------------------
# SAGAN attention core (libs/self_attention.py:56-63: s = matmul(f, g^T), beta = softmax(s), o = matmul(beta, h)) and the
# closed forms of its first- and second-order backward that csrc/flash.hip evaluates tile by tile.  The reference gets
# both from tf.gradients (the gradient penalty, image_generation.py:414-439, differentiates the first backward); these
# restatements are pinned against float64 autograd of the three ops in tests/test_oracle.py.
# ------------------------------------------------------------------------------------------------
def attention_forward(q, k, v):
  """q, k [n, N, dk], v [n, N, dv] -> (o, p): o = softmax(q k^T) v."""
  s = np.einsum('nid,njd->nij', q, k)
  p = np.exp(s - s.max(-1, keepdims=True))
  p /= p.sum(-1, keepdims=True)
  return np.einsum('nij,njd->nid', p, v), p


def attention_backward(q, k, v, go):
  """(dq, dk, dv) of <o, go>: gP = go v^T, D = rowsum(P o gP), gS = P o (gP - D)."""
  _, p = attention_forward(q, k, v)
  gp = np.einsum('nid,njd->nij', go, v)
  gs = p * (gp - (p * gp).sum(-1, keepdims=True))
  return np.einsum('nij,njd->nid', gs, k), np.einsum('nij,nid->njd', gs, q), np.einsum('nij,nid->njd', p, go)


def attention_backward_backward(q, k, v, go, aq, ak, av):
  """Gradients of <dq, aq> + <dk, ak> + <dv, av> with (dq, dk, dv) = attention_backward(q, k, v, go), with respect to
  q, k, v and go:  W = aq k^T + q ak^T, E = rowsum(P o W), T = P o (W - E);  Y = go av^T,
  X = Y + W o (gP - D) - E gP, F = rowsum(P o X), U = P o (X - F);
  adj q = gS ak + U k,  adj k = gS^T aq + U^T q,  adj v = T^T go,  adj go = P av + T v."""
  _, p = attention_forward(q, k, v)
  gp = np.einsum('nid,njd->nij', go, v)
  d = (p * gp).sum(-1, keepdims=True)
  gs = p * (gp - d)
  w = np.einsum('nid,njd->nij', aq, k) + np.einsum('nid,njd->nij', q, ak)
  e = (p * w).sum(-1, keepdims=True)
  t = p * (w - e)
  x = np.einsum('nid,njd->nij', go, av) + w * (gp - d) - e * gp
  u = p * (x - (p * x).sum(-1, keepdims=True))
  adj_q = np.einsum('nij,njd->nid', gs, ak) + np.einsum('nij,njd->nid', u, k)
  adj_k = np.einsum('nij,nid->njd', gs, aq) + np.einsum('nij,nid->njd', u, q)
  adj_v = np.einsum('nij,nid->njd', t, go)
  adj_go = np.einsum('nij,njd->nid', p, av) + np.einsum('nij,njd->nid', t, v)
  return adj_q, adj_k, adj_v, adj_go
