"""Input pipeline: TFRecord files of tf.Example protos -> decoded images -> the preprocessed [B, hw, hw, 3] batches the
trainer consumes, with the resize / flip / colour distortion on the GPU (csrc/preprocess.hip).

Reference call sites: datasets/image_only.py:45-106 (`get_split`: files `<dataset_dir>/<split>*`,
features image/encoded, image/format [default 'jpeg'], image/filename; slim.tfexample_decoder.Image),
datasets/dataset_utils.py:82-90,495-513 (the converters that write those records), model/model_inheritor.py:786-830
(DatasetDataProvider: shuffled parallel readers -> preprocessing threads -> tf.train.batch) and
preprocessing/danbooru_preprocessing.py:115-230 (preprocess_image).  TwinGAN trains on two such datasets, one per domain
(`--dataset_name` / `--unpaired_target_dataset_name`, twingan.py:146-200).

On-disk formats are restated from TensorFlow's published sources (no TensorFlow here; "parity unpinned" for the file
formats, like checkpoint.py): a TFRecord file is a sequence of [uint64 length | masked CRC-32C of the length | payload |
masked CRC-32C of the payload] (core/lib/io/record_writer.cc); tf.Example is the protobuf
Example{1: Features{1: map<string, Feature{1: BytesList | 2: FloatList | 3: Int64List}>}} (core/example/*.proto).
The preprocessing itself IS pinned: the reference's own preprocess_image runs on the TF stand-in
(oracle/ref_runner.run_preprocess) and the HIP kernel is held to it (tests/golden/preprocess_hw32.npz).
"""
import glob
import io
import os
import struct
import threading
import queue as _queue

import numpy as np
import torch

from . import _lib
from ._lib import TG_BF16, TG_F16, TG_F32, call
from .checkpoint import _field, _get_varint, _parse_message, _put_varint, crc32c, mask_crc


# ------------------------------------------------------------------------------------------------ TFRecord
def write_tfrecords(path, payloads):
  """Writes byte strings as one TFRecord file."""
  with open(path, 'wb') as fh:
    for data in payloads:
      head = struct.pack('<Q', len(data))
      fh.write(head + struct.pack('<I', mask_crc(crc32c(head))) + data + struct.pack('<I', mask_crc(crc32c(data))))


def read_tfrecords(path, verify=False):
  """Yields the payloads of a TFRecord file (``verify``: check both checksums of every record)."""
  with open(path, 'rb') as fh:
    while True:
      head = fh.read(12)
      if not head:
        return
      if len(head) < 12:
        raise ValueError('%s: truncated record header' % path)
      n, = struct.unpack('<Q', head[:8])
      if verify and struct.unpack('<I', head[8:])[0] != mask_crc(crc32c(head[:8])):
        raise ValueError('%s: corrupt record length' % path)
      body = fh.read(n + 4)
      if len(body) < n + 4:
        raise ValueError('%s: truncated record' % path)
      if verify and struct.unpack('<I', body[n:])[0] != mask_crc(crc32c(body[:n])):
        raise ValueError('%s: corrupt record payload' % path)
      yield body[:n]


# ------------------------------------------------------------------------------------------------ tf.Example
def encode_example(features):
  """{name: bytes | str | [bytes] | int | [int] | float | [float] | ndarray} -> serialized tf.Example."""
  entries = b''
  for name in sorted(features):
    v = features[name]
    if isinstance(v, (bytes, str)):
      v = [v]
    elif isinstance(v, (int, float, np.integer, np.floating)):
      v = [v]
    v = list(v)
    if v and isinstance(v[0], (bytes, str)):
      items = b''.join(_field(1, 2, _put_varint(len(b)) + b) for b in ((x.encode() if isinstance(x, str) else x) for x in v))
      feat = _field(1, 2, _put_varint(len(items)) + items)                       # bytes_list
    elif v and isinstance(v[0], (float, np.floating)):
      packed = struct.pack('<%df' % len(v), *v)
      lst = _field(1, 2, _put_varint(len(packed)) + packed)
      feat = _field(2, 2, _put_varint(len(lst)) + lst)                           # float_list (packed)
    else:
      packed = b''.join(_put_varint(int(x)) for x in v)
      lst = _field(1, 2, _put_varint(len(packed)) + packed)
      feat = _field(3, 2, _put_varint(len(lst)) + lst)                           # int64_list (packed)
    key = name.encode()
    entry = _field(1, 2, _put_varint(len(key)) + key) + _field(2, 2, _put_varint(len(feat)) + feat)
    entries += _field(1, 2, _put_varint(len(entry)) + entry)
  return _field(1, 2, _put_varint(len(entries)) + entries)


def decode_example(payload):
  """serialized tf.Example -> {name: [bytes] | [float] | [int]}."""
  out = {}
  for feats in _parse_message(payload).get(1, []):
    for entry in _parse_message(feats).get(1, []):
      e = _parse_message(entry)
      name = e[1][0].decode()
      feat = _parse_message(e.get(2, [b''])[0])
      if 1 in feat:
        out[name] = _parse_message(feat[1][0]).get(1, [])
      elif 2 in feat:
        vals = []
        for chunk in _parse_message(feat[2][0]).get(1, []):
          if isinstance(chunk, bytes):
            vals.extend(struct.unpack('<%df' % (len(chunk) // 4), chunk))
          else:                                                                  # unpacked: fixed32 bit pattern
            vals.append(struct.unpack('<f', struct.pack('<I', chunk))[0])
        out[name] = vals
      elif 3 in feat:
        vals = []
        for chunk in _parse_message(feat[3][0]).get(1, []):
          if isinstance(chunk, bytes):
            pos = 0
            while pos < len(chunk):
              v, pos = _get_varint(chunk, pos)
              vals.append(v - (1 << 64) if v >= (1 << 63) else v)
          else:
            vals.append(chunk - (1 << 64) if chunk >= (1 << 63) else chunk)
        out[name] = vals
      else:
        out[name] = []
  return out


def decode_image(encoded, fmt=b'jpeg', channels=3):
  """slim.tfexample_decoder.Image: 'raw' / 'RAW' -> the bytes are the pixels; otherwise decode_png / decode_jpeg by
  content.  -> uint8 [h, w, channels]."""
  if fmt in (b'raw', b'RAW'):
    raise ValueError('raw images carry no shape in this dataset')
  from PIL import Image
  im = Image.open(io.BytesIO(encoded))
  im = im.convert('RGB' if channels == 3 else 'L')
  a = np.asarray(im, dtype=np.uint8)
  return a if a.ndim == 3 else a[:, :, None]


def image_example(image_bytes, fmt='jpeg', filename=''):
  """The record the reference's converters write for an image-only dataset (dataset_utils.py:82-90)."""
  return encode_example({'image/encoded': image_bytes, 'image/format': fmt, 'image/filename': filename})


class ImageOnlyDataset:
  """datasets/image_only.get_split: the TFRecord files `<dataset_dir>/<split>*`, each record a tf.Example with
  image/encoded (+ image/format, image/filename)."""

  def __init__(self, dataset_dir, split='train', file_pattern='%s*', key='image/encoded'):
    self.files = sorted(glob.glob(os.path.join(dataset_dir, file_pattern % split)))
    if not self.files:
      raise FileNotFoundError('no files match %s' % os.path.join(dataset_dir, file_pattern % split))
    self.key = key

  def records(self, files=None):
    for f in (files or self.files):
      yield from read_tfrecords(f)

  def decode(self, payload):
    ex = decode_example(payload)
    fmt = ex.get('image/format', [b'jpeg'])
    name = ex.get('image/filename', [b''])
    return decode_image(ex[self.key][0], fmt[0] if fmt else b'jpeg'), (name[0].decode() if name else '')

  def __iter__(self):
    for rec in self.records():
      yield self.decode(rec)


EMBEDDING_SIZE = 512      # datasets/celeba_facenet.py:45 DEFAULT_ENCODING_SIZE


def embedding_example(image_bytes, embedding, fmt='jpeg', filename='', **more):
  """A record of an image + embedding dataset as datasets/celeba_facenet.py:86-99 reads it: image/encoded, image/format,
  image/filename and the float vector image/embedding (FaceNet's 512 numbers there); ``more``: further features
  (image/attribs, image/landmarks ... are decoded by the reference but not consumed by the TwinGAN trainer)."""
  feats = {'image/encoded': image_bytes, 'image/format': fmt, 'image/filename': filename,
           'image/embedding': [float(v) for v in np.asarray(embedding, np.float32).reshape(-1)]}
  feats.update(more)
  return encode_example(feats)


class EmbeddingImageDataset(ImageOnlyDataset):
  """datasets/celeba_facenet.get_split (:50-118; `--dataset_name celeba_facenet`, dataset_factory.py:50-58): images with a
  precomputed embedding per image -- the item 'embedding' that the trainer reads back as 'a_embedding' / 'b_embedding'
  for --do_encoder_distillation (twingan.py:103,162-177).  decode() -> (image, filename, {'embedding': fp32 [size]});
  ``embedding_size`` = the FixedLenFeature's length (a record of another length is an error, as in tf.parse_example)."""
  fields = ('embedding',)

  def __init__(self, dataset_dir, split='train', file_pattern='%s-*', key='image/encoded', embedding_size=EMBEDDING_SIZE):
    ImageOnlyDataset.__init__(self, dataset_dir, split, file_pattern, key)
    self.embedding_size = int(embedding_size)

  def decode(self, payload):
    ex = decode_example(payload)
    fmt = ex.get('image/format', [b'jpeg'])
    name = ex.get('image/filename', [b''])
    emb = np.asarray(ex.get('image/embedding', ()), np.float32)
    if emb.shape != (self.embedding_size,):
      raise ValueError('image/embedding: expected %d floats, the record holds %d' % (self.embedding_size, emb.size))
    return (decode_image(ex[self.key][0], fmt[0] if fmt else b'jpeg'), (name[0].decode() if name else ''),
            {'embedding': emb})


def get_dataset(name, split, dataset_dir, embedding_size=EMBEDDING_SIZE):
  """datasets/dataset_factory.get_dataset (:61-91) for what the TwinGAN trainer consumes of each dataset: the decoded
  image ('a_source' / 'b_source', twingan.py:150-152) and, where the dataset has one, the 'embedding' item.  'anime_faces'
  (the target domain of the reference's own training recipe, docs/training.md:16-17) and 'celeba' carry labels / tags /
  landmarks as further features: the trainer never reads them ('conditional_labels' only feeds the conditional
  generators of image_generation.py's other programs), so they are skipped here -- the image key, format key and file
  pattern ('<split>-*', datasets/anime_faces.py:30, celeba.py:32) are theirs."""
  if name == 'image_only':
    return ImageOnlyDataset(dataset_dir, split)                                   # datasets/image_only.py:28 '%s*'
  if name in ('anime_faces', 'celeba'):
    return ImageOnlyDataset(dataset_dir, split, file_pattern='%s-*')
  if name == 'celeba_facenet':
    return EmbeddingImageDataset(dataset_dir, split, embedding_size=embedding_size)
  raise ValueError('Name of dataset unknown %s' % name)                           # dataset_factory.py:77-78


DATASETS = ('image_only', 'anime_faces', 'celeba', 'celeba_facenet')      # of datasets/dataset_factory.py:50-58


# ------------------------------------------------------------------------------------------------ GPU preprocessing
def source_rect(h, w, resize_mode, new_hw=None, rng=None, offset=None):
  """(y0, x0, sh, sw): the rectangle of preprocessing_util.resize_image (:97-146) in image coordinates that the first
  resize reads (see twingan_hip.h).  ``new_hw``: that resize's target (hw, or int(hw / ratio) with random cropping) --
  RANDOM_CROP (:84-95,126-127,144-146) cuts a [new_hw, new_hw] window at a uniform offset (``rng``) out of an image at
  least that large, which the resize then copies 1:1, and resizes a smaller image whole; NONE (:137-139) takes the image
  as it is, so it must already have the target size.  ``offset`` = (oy, ox): the window's position instead of a draw (tests)."""
  if resize_mode == 'PAD':
    size = max(h, w)
    return (-((size - h) // 2), -((size - w) // 2), size, size)
  if resize_mode == 'CROP':
    size = min(h, w)
    return ((h - size) // 2, (w - size) // 2, size, size)
  if resize_mode == 'RESHAPE':
    return (0, 0, h, w)
  if resize_mode in ('RANDOM_CROP', 'RANDOM_CROP_AND_RESHAPE'):      # for the latter new_hw is the initial crop size
    if new_hw > min(h, w):
      return (0, 0, h, w)
    oy, ox = offset if offset is not None else (rng.integers(0, h - new_hw + 1), rng.integers(0, w - new_hw + 1))
    assert 0 <= oy <= h - new_hw and 0 <= ox <= w - new_hw
    return (int(oy), int(ox), new_hw, new_hw)
  if resize_mode == 'NONE':
    if (h, w) != (new_hw, new_hw):
      raise ValueError('resize_mode NONE: a %d x %d image where the networks take %d x %d' % (h, w, new_hw, new_hw))
    return (0, 0, h, w)
  raise ValueError('resize_mode %s (PAD, CROP, RESHAPE, RANDOM_CROP, RANDOM_CROP_AND_RESHAPE and NONE)' % resize_mode)


def draw_augmentation(n, rng):
  """The random draws of preprocess_image(is_training=True) for n images -> fp32 [n, 4] (flip, saturation first,
  brightness delta, saturation factor): random_flip_left_right flips when U[0,1) < 0.5; apply_with_random_selector picks
  one of 4 orderings, of which only ordering 0 puts brightness first in fast mode; random_brightness(32/255),
  random_saturation(0.5, 1.5) (danbooru_preprocessing.py:62-113)."""
  aug = np.empty((n, 4), np.float32)
  aug[:, 0] = rng.random(n) < 0.5
  aug[:, 1] = rng.integers(0, 4, n) != 0
  aug[:, 2] = rng.uniform(-32.0 / 255.0, 32.0 / 255.0, n)
  aug[:, 3] = rng.uniform(0.5, 1.5, n)
  return aug


RANDOM_CROP_RATIO = 0.8                                        # danbooru_preprocessing.py:33
COLOR_SPACES = {'rgb': 0, 'yiq': 1, 'bgr': 2, 'gray': 3}       # danbooru_preprocessing.py:31; TG_CS_* of preprocess.hip


def draw_crops(n, mid, ratio, rng):
  """The draws of preprocessing_util.random_crop_image (:312-331) on the [mid, mid] image for n images -> int32 [n, 4] =
  (cy, cx, ch, cw): ch = int32(mid * U[ratio, 1)), cw likewise from its own draw (the products are formed in float32, as
  the graph does, and truncated), then tf.random_crop's offsets, uniform over [0, mid - size]."""
  crop = np.empty((n, 4), np.int32)
  for col in (2, 3):
    u = np.minimum(rng.uniform(ratio, 1.0, n).astype(np.float32), np.nextafter(np.float32(1.0), np.float32(0.0)))
    crop[:, col] = (np.float32(mid) * u).astype(np.int32)
  crop[:, 0] = rng.integers(0, mid - crop[:, 2] + 1)
  crop[:, 1] = rng.integers(0, mid - crop[:, 3] + 1)
  return crop


class Preprocessor:
  """preprocess_image for a batch of decoded images on the GPU.  ``aug`` (and ``crop``) given explicitly (tests) or
  drawn from ``rng``; evaluation (is_training=False) resizes only.  ``do_random_cropping`` / ``color_space``: the flags of
  model_inheritor.py:225,240 (the reference's training recipe, docs/training.md:22-23, is --resize_mode=RESHAPE
  --do_random_cropping=True)."""

  def __init__(self, hw, device='cuda', precision='bf16', resize_mode='PAD', is_training=True, seed=0,
               do_random_cropping=False, random_cropping_ratio=RANDOM_CROP_RATIO, color_space='rgb', initial_crop_hw=None):
    self.hw, self.device = int(hw), torch.device(device)
    self.dtype = {'bf16': torch.bfloat16, 'fp16': torch.float16, 'fp32': torch.float32}[precision]
    self.resize_mode, self.is_training = resize_mode, is_training
    assert color_space in COLOR_SPACES, 'color_space must be one of %s' % sorted(COLOR_SPACES)      # _check_color_space
    self.color_space = color_space
    # danbooru_preprocessing.py:187-190: only a TRAINING call crops; the first resize then goes to int(hw / ratio)
    self.crops = bool(do_random_cropping and is_training)
    assert not (self.crops and resize_mode == 'NONE'), 'random cropping of unresized images is not built'
    self.ratio = float(random_cropping_ratio)
    self.mid = int(self.hw / self.ratio) if self.crops else 0
    # RANDOM_CROP_AND_RESHAPE (preprocessing_util.py:24-27,128-131; --random_crop_and_reshape_initial_crop_hw): a random
    # [c, c] window of the image (a smaller image is resized up to [c, c] first), then the resize to hw -- the kernel's
    # two-stage path with the intermediate size c and the whole intermediate image as its "crop", in evaluation as well
    # (resize_image does not look at is_training).  Together with --do_random_cropping it would be three resizes: not built.
    self.window = None
    if resize_mode == 'RANDOM_CROP_AND_RESHAPE':
      assert initial_crop_hw and int(initial_crop_hw) > 0, 'RANDOM_CROP_AND_RESHAPE needs initial_crop_hw'      # :129
      assert not self.crops, 'RANDOM_CROP_AND_RESHAPE together with do_random_cropping (three resizes) is not built'
      self.window = self.mid = int(initial_crop_hw)
      # tg_preprocess_images_crop takes an intermediate image at least as large as its output (the cropping augmentation
      # always shrinks); a window SMALLER than the training resolution (an up-sampling crop) is refused here
      assert self.window >= self.hw, 'initial_crop_hw %d < hw %d: up-sampling windows are not built' % (self.window, self.hw)
    self.rng = np.random.default_rng(seed)

  def pack(self, images, aug=None, crop=None, rng=None, mode_offsets=None):
    """Host side: one pinned uint8 buffer + the per-image tables (+ the crop table when cropping is on).  ``rng``: the
    caller's random stream (loader workers own one each); default: the preprocessor's.  ``mode_offsets``: per image the
    RANDOM_CROP window's (oy, ox) or None (tests feed the reference's draws)."""
    rng = rng or self.rng
    n = len(images)
    sizes = [int(im.shape[0]) * int(im.shape[1]) * 3 for im in images]
    offsets = np.zeros(n, np.int64)
    offsets[1:] = np.cumsum(sizes[:-1])
    buf = torch.empty(int(sum(sizes)), dtype=torch.uint8, pin_memory=self.device.type == 'cuda')
    flat = buf.numpy()
    rect = np.empty((n, 6), np.int32)
    for i, im in enumerate(images):
      assert im.dtype == np.uint8 and im.ndim == 3 and im.shape[2] == 3, 'decoded RGB uint8 images'
      flat[offsets[i]:offsets[i] + sizes[i]] = np.ascontiguousarray(im).reshape(-1)
      rect[i] = (im.shape[0], im.shape[1]) + source_rect(im.shape[0], im.shape[1], self.resize_mode,
                                                         self.mid if (self.crops or self.window is not None) else self.hw, rng,
                                                         None if mode_offsets is None else mode_offsets[i])
    if aug is None:
      aug = draw_augmentation(n, rng) if self.is_training else np.tile(np.float32([0, 0, 0, 1]), (n, 1))
    tables = (buf, torch.from_numpy(offsets), torch.from_numpy(rect), torch.from_numpy(np.ascontiguousarray(aug, np.float32)))
    if self.window is not None:      # the second stage reads the whole [c, c] intermediate image
      assert crop is None
      return tables + (torch.from_numpy(np.tile(np.int32([0, 0, self.window, self.window]), (n, 1))),)
    if not self.crops:
      assert crop is None, 'a crop table without do_random_cropping (or in evaluation mode)'
      return tables
    if crop is None:
      crop = draw_crops(n, self.mid, self.ratio, rng)
    crop = np.ascontiguousarray(crop, np.int32).reshape(n, 4)
    assert (crop[:, :2] >= 0).all() and (crop[:, 2:] >= 1).all() and (crop[:, :2] + crop[:, 2:] <= self.mid).all(), \
        'crop rectangles must lie inside the %d x %d intermediate image' % (self.mid, self.mid)
    return tables + (torch.from_numpy(crop),)

  def __call__(self, images, aug=None, stream=None, crop=None, mode_offsets=None):
    return self.run(*self.pack(images, aug, crop, mode_offsets=mode_offsets), stream=stream)

  def run(self, buf, offsets, rect, aug, crop=None, stream=None):
    if self.device.type != 'cuda':
      raise RuntimeError('the preprocessing kernel needs a GPU (there is no CPU fallback)')
    n = offsets.numel()
    two_stage = self.crops or self.window is not None
    assert (crop is not None) == two_stage
    with torch.cuda.device(self.device):
      st = stream or torch.cuda.current_stream()
      with torch.cuda.stream(st):
        d = [t.to(self.device, non_blocking=True) for t in (buf, offsets, rect, aug) + ((crop,) if two_stage else ())]
        out = torch.empty((n, self.hw, self.hw, 3), dtype=self.dtype, device=self.device)
        call('tg_preprocess_images_crop', d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(),
             d[4].data_ptr() if two_stage else 0, d[3].data_ptr(), out.data_ptr(), n, self.hw, self.mid,
             COLOR_SPACES[self.color_space], {torch.bfloat16: TG_BF16, torch.float16: TG_F16, torch.float32: TG_F32}[self.dtype],
             st.cuda_stream)
        for t in d:
          t.record_stream(st)
    return out


# ------------------------------------------------------------------------------------------------ loader
def _stack_fields(decoded):
  """The per-image field dictionaries of a batch (EmbeddingImageDataset.decode()[2]) -> {name: fp32 tensor [batch, size]};
  None for an image-only dataset."""
  if len(decoded[0]) < 3:
    return None
  return {k: torch.from_numpy(np.stack([d[2][k] for d in decoded])) for k in decoded[0][2]}


def _process_main(files, key, batch_size, hw, resize_mode, is_training, shuffle, pool_size, seed, out_q, stop,
                  crop_kw=None, ds_state=None):
  """A decode worker PROCESS of Loader(processes=P): its own file shard, shuffling pool and random stream; puts packed
  batches (shared-memory tensors) on ``out_q``."""
  torch.set_num_threads(1)
  ds_cls, ds_attrs = ds_state or (ImageOnlyDataset, {})
  ds = ds_cls.__new__(ds_cls)
  ds.files, ds.key = list(files), key
  ds.__dict__.update(ds_attrs)
  pre = Preprocessor(hw, device='cpu', resize_mode=resize_mode, is_training=is_training, **(crop_kw or {}))
  rng = np.random.default_rng(seed)
  held = []
  want = max(1, pool_size) if shuffle else 1
  images = []

  def records():
    while True:
      order = list(files)
      if shuffle:
        rng.shuffle(order)
      yield from ds.records(order)
  for rec in records():
    if stop.is_set():
      return
    held.append(rec)
    if len(held) < want:
      continue
    k = int(rng.integers(len(held))) if shuffle else 0
    held[k], held[-1] = held[-1], held[k]
    images.append(ds.decode(held.pop()))
    if len(images) == batch_size:
      fields = _stack_fields(images)
      item = (tuple(t.share_memory_() for t in pre.pack([d[0] for d in images], rng=rng)),
              None if fields is None else {k: v.share_memory_() for k, v in fields.items()})
      images = []
      while not stop.is_set():
        try:
          out_q.put(item, timeout=0.1)
          break
        except _queue.Full:
          continue


class Loader:
  """DatasetDataProvider + tf.train.batch (model_inheritor.py:786-830, 380-400): ``num_readers`` reader threads walk
  shuffled file lists forever and push records into a shuffling pool; ``num_workers`` threads decode (PIL releases the
  GIL inside libjpeg) and assemble packed batches; the consumer uploads and preprocesses on a side stream while the
  previous batch trains.  next() -> device tensor [batch, hw, hw, 3].
  ``processes`` = P > 0: decoding in P worker processes instead (threads top out near 3.5 k images/s on the
  interpreter lock: tf.Example parsing and the array copies hold it), each with its own file shard, shuffling pool
  (pool / P records) and random stream, handing over packed batches in shared memory."""

  def __init__(self, dataset, batch_size, preprocessor, num_readers=4, num_workers=8, shuffle=True, pool=None, seed=0,
               prefetch=4, processes=0, stall_timeout=600.0):
    """``stall_timeout``: seconds next() waits for a batch while every decode process is alive before it raises
    (None / 0: wait for ever)."""
    self.ds, self.bs, self.pre = dataset, int(batch_size), preprocessor
    self.stall_timeout = stall_timeout
    self.shuffle = shuffle
    self.pool_size = pool if pool is not None else 20 * self.bs      # common_queue_capacity = 20 * batch_size
    self.stream = torch.cuda.Stream(device=preprocessor.device) if preprocessor.device.type == 'cuda' else None
    self.threads, self.procs = [], []
    if processes > 0:
      import torch.multiprocessing as mp
      ctx = mp.get_context('spawn')
      self.stop = ctx.Event()
      self.batches = ctx.Queue(maxsize=max(prefetch, 2 * processes))
      files = list(dataset.files)
      for r in range(processes):
        mine = files[r::processes] or files
        pr = ctx.Process(target=_process_main, daemon=True,
                         args=(mine, dataset.key, self.bs, preprocessor.hw, preprocessor.resize_mode,
                               preprocessor.is_training, shuffle, max(1, self.pool_size // processes), seed + 1 + r,
                               self.batches, self.stop,
                               dict(do_random_cropping=preprocessor.crops, random_cropping_ratio=preprocessor.ratio,
                                    initial_crop_hw=preprocessor.window),
                               (type(dataset), {k: v for k, v in dataset.__dict__.items() if k not in ('files', 'key')})))
        pr.start()
        self.procs.append(pr)
      return
    self.records = _queue.Queue(maxsize=self.pool_size)
    self.batches = _queue.Queue(maxsize=prefetch)
    self.stop = threading.Event()
    self.rng = np.random.default_rng(seed)
    files = list(dataset.files)
    for r in range(num_readers):
      mine = files[r::num_readers] or files
      self.threads.append(threading.Thread(target=self._read, args=(mine, seed + 1 + r), daemon=True))
    for w in range(num_workers):
      self.threads.append(threading.Thread(target=self._work, args=(seed + 1000 + w,), daemon=True))
    for t in self.threads:
      t.start()

  def _read(self, files, seed):
    rng = np.random.default_rng(seed)
    while not self.stop.is_set():
      order = list(files)
      if self.shuffle:
        rng.shuffle(order)
      for rec in self.ds.records(order):
        while not self.stop.is_set():
          try:
            self.records.put(rec, timeout=0.1)
            break
          except _queue.Full:
            continue
        if self.stop.is_set():
          return

  def _work(self, seed):
    rng = np.random.default_rng(seed)
    held = []                                   # a private shuffling pool (RandomShuffleQueue semantics)
    want = max(1, self.pool_size // 8) if self.shuffle else 1
    while not self.stop.is_set():
      images = []
      while len(images) < self.bs and not self.stop.is_set():
        try:
          held.append(self.records.get(timeout=0.1))
        except _queue.Empty:
          continue
        if len(held) < want:
          continue
        k = int(rng.integers(len(held))) if self.shuffle else 0
        held[k], held[-1] = held[-1], held[k]
        images.append(self.ds.decode(held.pop()))
      if len(images) == self.bs:
        packed = (self.pre.pack([d[0] for d in images], rng=rng), _stack_fields(images))
        while not self.stop.is_set():
          try:
            self.batches.put(packed, timeout=0.1)
            break
          except _queue.Full:
            continue

  def next(self):
    """-> images [batch, hw, hw, 3] on the device; for a dataset with further fields (EmbeddingImageDataset):
    (images, {field: fp32 device tensor [batch, size]})."""
    if self.procs:      # a worker process that died (bad record, failed assert) must raise here, not hang the consumer
      waited = 0.0
      while True:
        try:
          packed, fields = self.batches.get(timeout=1.0)
          break
        except _queue.Empty as e:      # nothing queued: a producer that is gone, or workers that are alive but stuck
          waited += 1.0
          dead = [pr for pr in self.procs if not pr.is_alive()]
          if dead:      # workers loop until close(): an exit -- exit code 0 included -- before that is a failure
            raise RuntimeError('Loader: %d of %d decode worker process(es) exited (exit codes %s)'
                               % (len(dead), len(self.procs), [pr.exitcode for pr in dead])) from e
          if self.stall_timeout and waited >= self.stall_timeout:
            raise RuntimeError('Loader: no batch for %.0f s with all %d decode workers alive (stall_timeout)'
                               % (waited, len(self.procs))) from e
        except Exception as e:      # a batch whose shared memory went away with its (dead) producer, an unpickling error
          dead = [pr for pr in self.procs if not pr.is_alive()]
          if dead:
            raise RuntimeError('Loader: %d of %d decode worker process(es) exited (exit codes %s)'
                               % (len(dead), len(self.procs), [pr.exitcode for pr in dead])) from e
          raise
    else:
      packed, fields = self.batches.get()
    out = self.pre.run(*packed, stream=self.stream)
    if fields is not None:
      with torch.cuda.stream(self.stream):
        fields = {k: v.to(self.pre.device, non_blocking=True) for k, v in fields.items()}
    if self.stream is not None:
      cur = torch.cuda.current_stream(self.pre.device)
      cur.wait_stream(self.stream)
      out.record_stream(cur)
      for v in (fields or {}).values():
        v.record_stream(cur)
    return out if fields is None else (out, fields)

  def close(self):
    self.stop.set()
    for t in self.threads:
      t.join(timeout=2.0)
    for pr in self.procs:
      pr.join(timeout=2.0)
      if pr.is_alive():
        pr.terminate()
    if self.procs:      # drop what is left in the queue so that its feeder threads can exit
      try:
        while True:
          self.batches.get_nowait()
      except Exception:
        pass


class TwoDomainBatches:
  """The ``batch_fn`` of runner.run_progressive over two image-only datasets -- TwinGAN's unpaired source / target
  domains (twingan.py:146-200: `--dataset_name` and `--unpaired_target_dataset_name`, one provider each, both through
  preprocess_image at the stage's resolution).  A loader pair is (re)built whenever the stage's resolution or batch size
  changes, as the reference rebuilds its graph per stage."""

  def __init__(self, source_dir, target_dir, device='cuda', precision='bf16', split='train', resize_mode='PAD',
               processes=0, num_workers=8, seed=0, do_random_cropping=False, color_space='rgb',
               dataset_names=('image_only', 'image_only'), embedding_size=EMBEDDING_SIZE):
    self.dirs = (source_dir, target_dir)
    self.dataset_names, self.embedding_size = tuple(dataset_names), embedding_size
    self.kw = dict(device=device, precision=precision, resize_mode=resize_mode, do_random_cropping=do_random_cropping,
                   color_space=color_space)
    self.split, self.processes, self.num_workers, self.seed = split, processes, num_workers, seed
    self.key, self.loaders = None, ()

  def __call__(self, hw, batch_size):
    if self.key != (hw, batch_size):
      self.close()
      self.loaders = tuple(
          Loader(self._dataset(i, d), batch_size, Preprocessor(hw, seed=self.seed + 17 * i, **self.kw),
                 num_workers=self.num_workers, processes=self.processes, seed=self.seed + 1000 * i)
          for i, d in enumerate(self.dirs))
      self.key = (hw, batch_size)
    s, t = self.loaders[0].next(), self.loaders[1].next()
    extras = {}
    # the provider's 'a_embedding' / 'b_embedding' (twingan.py:103,162-177): whichever dataset carries embeddings
    if isinstance(s, tuple):
      s, extras['distill_embed_s'] = s[0], s[1]['embedding']
    if isinstance(t, tuple):
      t, extras['distill_embed_t'] = t[0], t[1]['embedding']
    return (s, t, extras) if extras else (s, t)

  def _dataset(self, i, d):
    return get_dataset(self.dataset_names[i], self.split, d, self.embedding_size)

  def close(self):
    for ld in self.loaders:
      ld.close()
    self.loaders = ()
