"""TF-name-compatible checkpoint I/O: the TensorFlow-1.x "V2" checkpoint (tensor bundle) read and written in pure
Python + NumPy, and the warm-start rule of the reference on top of it.

Reference call sites: the trainer saves through ``tf.train.Saver`` (slim.learning.train, model/model_inheritor.py:
537-571 -> files ``<train_dir>/model.ckpt-<global_step>.{index,data-00000-of-00001}`` + the text file ``checkpoint``);
a stage warm-starts from the previous stage's directory with ``_get_init_fn`` (model/model_inheritor.py:576-644:
model variables only, ``checkpoint_exclude_scopes``, ``tf.train.latest_checkpoint`` of a directory,
``ignore_missing_vars``; pggan_runner.py:136-146 sets the flags); inference restores the same files
(inference/image_translation_infer.py:46-99).  Variables keep TensorFlow's names and layouts here (params.py, SURVEY.md
Appendix C), so a checkpoint is a name -> array dictionary in both directions.

Format (restated from TensorFlow's published sources; tensorflow==1.8 is a requirement of the reference that is not
installable here -- requirement.txt:1 -- so this module is checked against its own writer, the format's published test
vectors (CRC-32C, varints) and hand-assembled files, NOT against a file written by TensorFlow: "parity unpinned" for
this row):
  * ``<prefix>.index`` is a LevelDB-format sorted table (tensorflow/core/lib/io/table*.cc, format.cc): data blocks of
    prefix-compressed entries [shared varint32 | non_shared varint32 | value_len varint32 | key suffix | value] with a
    restart array [uint32 offsets..., uint32 count]; every block is followed by a 1-byte compression type (0 none,
    1 snappy) and a masked CRC-32C of block + type; an index block maps separator keys to BlockHandles (offset, size as
    varint64); the 48-byte footer holds the metaindex and index handles and the magic 0xdb4775248b80fb57.
  * key "" -> BundleHeaderProto {1: num_shards, 2: endianness, 3: VersionDef}; key <variable name> -> BundleEntryProto
    {1: dtype, 2: TensorShapeProto {2: dim {1: size}}, 3: shard_id, 4: offset, 5: size, 6: masked crc32c (fixed32)}
    (tensorflow/core/protobuf/tensor_bundle.proto, util/tensor_bundle/tensor_bundle.cc).
  * ``<prefix>.data-0000S-of-0000N``: the tensors' little-endian bytes at [offset, offset + size).
"""
import os
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
MASK_DELTA = 0xa282ead8

# tensorflow/core/framework/types.proto
DT_FLOAT, DT_DOUBLE, DT_INT32, DT_UINT8, DT_INT16, DT_INT8, DT_INT64, DT_BOOL, DT_BFLOAT16, DT_HALF = 1, 2, 3, 4, 5, 6, 9, 10, 14, 19
_NP_OF_DT = {DT_FLOAT: np.dtype('<f4'), DT_DOUBLE: np.dtype('<f8'), DT_INT32: np.dtype('<i4'), DT_UINT8: np.dtype('u1'),
             DT_INT16: np.dtype('<i2'), DT_INT8: np.dtype('i1'), DT_INT64: np.dtype('<i8'), DT_BOOL: np.dtype('?'),
             DT_HALF: np.dtype('<f2'), DT_BFLOAT16: np.dtype('<u2')}      # bfloat16 is returned as its uint16 bit pattern
_DT_OF_NP = {np.dtype('float32'): DT_FLOAT, np.dtype('float64'): DT_DOUBLE, np.dtype('int32'): DT_INT32,
             np.dtype('uint8'): DT_UINT8, np.dtype('int16'): DT_INT16, np.dtype('int8'): DT_INT8,
             np.dtype('int64'): DT_INT64, np.dtype('bool'): DT_BOOL, np.dtype('float16'): DT_HALF}


# ------------------------------------------------------------------------------------------------ CRC-32C (Castagnoli)
def _make_table():
  poly = 0x82f63b78
  t = np.zeros(256, np.uint32)
  for i in range(256):
    c = i
    for _ in range(8):
      c = (c >> 1) ^ (poly if c & 1 else 0)
    t[i] = c
  return t


_TABLE = _make_table()
_TABLE_LIST = [int(v) for v in _TABLE]


def _crc_bytes(state, data):
  """Raw (pre-/post-inversion NOT applied) CRC register after ``data``, byte at a time."""
  t = _TABLE_LIST
  for b in data:
    state = t[(state ^ b) & 0xff] ^ (state >> 8)
  return state


def _gf2_times(mat, vec):
  s, i = 0, 0
  while vec:
    if vec & 1:
      s ^= mat[i]
    vec >>= 1
    i += 1
  return s


def _gf2_square(mat):
  return [_gf2_times(mat, mat[n]) for n in range(32)]


def _shift_operator(nbytes):
  """The 32x32 GF(2) matrix (as 32 column words) that advances a raw CRC register over ``nbytes`` zero bytes."""
  odd = [0x82f63b78] + [1 << n for n in range(31)]      # one zero BIT
  op = [1 << n for n in range(32)]                        # identity
  bits = nbytes * 8
  sq = odd
  while bits:
    if bits & 1:
      op = [_gf2_times(sq, op[n]) for n in range(32)]
    bits >>= 1
    if bits:
      sq = _gf2_square(sq)
  return op


def crc32c(data):
  """CRC-32C of a bytes-like object.  Large buffers are cut into equal lanes whose registers advance in lockstep as
  NumPy vectors (the byte-serial table algorithm, vectorised ACROSS lanes); lane results are chained with the GF(2)
  operator that advances a register over one lane's length of zeros."""
  buf = np.frombuffer(memoryview(data).cast('B'), dtype=np.uint8)
  n = buf.size
  lanes = 1024
  if n < 64 * lanes:
    return _crc_bytes(0xffffffff, buf.tobytes()) ^ 0xffffffff
  step = n // lanes
  body = buf[:step * lanes].reshape(lanes, step)
  reg = np.zeros(lanes, np.uint32)
  reg[0] = 0xffffffff                                    # only the first lane carries the initial value
  for j in range(step):
    reg = _TABLE[(reg ^ body[:, j]) & 0xff] ^ (reg >> 8)
  op = _shift_operator(step)
  state = 0
  for r in reg.tolist():                                 # state = shift(state, step) ^ lane register (CRC is linear)
    state = _gf2_times(op, state) ^ r
  state = _crc_bytes(state, buf[step * lanes:].tobytes())
  return state ^ 0xffffffff


def mask_crc(crc):
  return ((((crc >> 15) | (crc << 17)) & 0xffffffff) + MASK_DELTA) & 0xffffffff


def unmask_crc(masked):
  rot = (masked - MASK_DELTA) & 0xffffffff
  return ((rot >> 17) | (rot << 15)) & 0xffffffff


# ------------------------------------------------------------------------------------------------ varints / protobuf wire
def _put_varint(v):
  out = bytearray()
  v &= (1 << 64) - 1
  while v >= 0x80:
    out.append((v & 0x7f) | 0x80)
    v >>= 7
  out.append(v)
  return bytes(out)


def _get_varint(buf, pos):
  shift = result = 0
  while True:
    b = buf[pos]
    pos += 1
    result |= (b & 0x7f) << shift
    if not b & 0x80:
      return result, pos
    shift += 7
    if shift > 63:
      raise ValueError('varint too long')


def _parse_message(buf):
  """{field number: [values]} of one protobuf message (varint -> int, length-delimited -> bytes, fixed32/64 -> int)."""
  out, pos = {}, 0
  while pos < len(buf):
    key, pos = _get_varint(buf, pos)
    field, wire = key >> 3, key & 7
    if wire == 0:
      v, pos = _get_varint(buf, pos)
    elif wire == 1:
      v = struct.unpack_from('<Q', buf, pos)[0]
      pos += 8
    elif wire == 2:
      ln, pos = _get_varint(buf, pos)
      v = bytes(buf[pos:pos + ln])
      pos += ln
    elif wire == 5:
      v = struct.unpack_from('<I', buf, pos)[0]
      pos += 4
    else:
      raise ValueError('unsupported protobuf wire type %d' % wire)
    out.setdefault(field, []).append(v)
  return out


def _field(num, wire, payload):
  return _put_varint((num << 3) | wire) + payload


def _entry_proto(dtype, shape, shard_id, offset, size, crc_masked):
  dims = b''
  for extent in shape:
    dim = _field(1, 0, _put_varint(int(extent)))                      # TensorShapeProto.Dim.size
    dims += _field(2, 2, _put_varint(len(dim)) + dim)                 # TensorShapeProto.dim
  msg = _field(1, 0, _put_varint(dtype)) + _field(2, 2, _put_varint(len(dims)) + dims)
  if shard_id:
    msg += _field(3, 0, _put_varint(shard_id))
  if offset:
    msg += _field(4, 0, _put_varint(offset))
  msg += _field(5, 0, _put_varint(size)) + _field(6, 5, struct.pack('<I', crc_masked))
  return msg


def _signed64(v):
  return v - (1 << 64) if v >= (1 << 63) else v


def _parse_entry(buf):
  m = _parse_message(buf)
  shape = []
  for sp in m.get(2, []):
    for dim in _parse_message(sp).get(2, []):
      shape.append(_signed64(_parse_message(dim).get(1, [0])[0]))
  if 7 in m:
    raise NotImplementedError('sliced (partitioned) variables are not supported')
  return dict(dtype=m.get(1, [0])[0], shape=tuple(shape), shard_id=m.get(3, [0])[0], offset=m.get(4, [0])[0],
              size=m.get(5, [0])[0], crc32c=m.get(6, [None])[0])


# ------------------------------------------------------------------------------------------------ snappy (read side)
def _snappy_uncompress(src):
  n, pos = _get_varint(src, 0)
  out = bytearray()
  while pos < len(src):
    tag = src[pos]
    pos += 1
    kind = tag & 3
    if kind == 0:                                          # literal
      ln = tag >> 2
      if ln >= 60:
        nb = ln - 59
        ln = int.from_bytes(src[pos:pos + nb], 'little')
        pos += nb
      ln += 1
      out += src[pos:pos + ln]
      pos += ln
      continue
    if kind == 1:
      ln = ((tag >> 2) & 7) + 4
      off = ((tag >> 5) << 8) | src[pos]
      pos += 1
    elif kind == 2:
      ln = (tag >> 2) + 1
      off = int.from_bytes(src[pos:pos + 2], 'little')
      pos += 2
    else:
      ln = (tag >> 2) + 1
      off = int.from_bytes(src[pos:pos + 4], 'little')
      pos += 4
    if off == 0 or off > len(out):
      raise ValueError('corrupt snappy block')
    for _ in range(ln):                                    # copies may overlap their own output
      out.append(out[-off])
  if len(out) != n:
    raise ValueError('corrupt snappy block: %d bytes, header says %d' % (len(out), n))
  return bytes(out)


# ------------------------------------------------------------------------------------------------ sorted table
def _read_block(data, offset, size, verify):
  raw = data[offset:offset + size]
  ctype = data[offset + size]
  if verify:
    want = struct.unpack_from('<I', data, offset + size + 1)[0]
    got = mask_crc(crc32c(data[offset:offset + size + 1]))
    if want != got:
      raise ValueError('table block at %d: checksum mismatch' % offset)
  if ctype == 0:
    return bytes(raw)
  if ctype == 1:
    return _snappy_uncompress(bytes(raw))
  raise ValueError('unknown block compression %d' % ctype)


def _block_entries(block):
  nrestarts = struct.unpack_from('<I', block, len(block) - 4)[0]
  limit = len(block) - 4 - 4 * nrestarts
  pos, key = 0, b''
  while pos < limit:
    shared, pos = _get_varint(block, pos)
    non_shared, pos = _get_varint(block, pos)
    vlen, pos = _get_varint(block, pos)
    key = key[:shared] + block[pos:pos + non_shared]
    pos += non_shared
    yield key, block[pos:pos + vlen]
    pos += vlen


def read_table(path, verify=True):
  """[(key bytes, value bytes)] of a LevelDB-format table file, in key order."""
  with open(path, 'rb') as fh:
    data = fh.read()
  if len(data) < 48 or struct.unpack_from('<Q', data, len(data) - 8)[0] != TABLE_MAGIC:
    raise ValueError('%s is not a table file (bad magic)' % path)
  footer = data[-48:]
  _, p = _get_varint(footer, 0)           # metaindex handle (unused)
  _, p = _get_varint(footer, p)
  ioff, p = _get_varint(footer, p)
  isize, p = _get_varint(footer, p)
  out = []
  for _, handle in _block_entries(_read_block(data, ioff, isize, verify)):
    boff, q = _get_varint(handle, 0)
    bsize, q = _get_varint(handle, q)
    out.extend(_block_entries(_read_block(data, boff, bsize, verify)))
  return out


class _BlockBuilder:
  def __init__(self, restart_interval=16):
    self.buf = bytearray()
    self.restarts = [0]
    self.count = 0
    self.last = b''
    self.interval = restart_interval

  def add(self, key, value):
    shared = 0
    if self.count < self.interval:
      m = min(len(key), len(self.last))
      while shared < m and key[shared] == self.last[shared]:
        shared += 1
    else:
      self.restarts.append(len(self.buf))
      self.count = 0
    self.buf += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value)) + key[shared:] + value
    self.last = key
    self.count += 1

  def finish(self):
    return bytes(self.buf) + b''.join(struct.pack('<I', r) for r in self.restarts) + struct.pack('<I', len(self.restarts))

  def size(self):
    return len(self.buf) + 4 * len(self.restarts) + 4


def write_table(path, items, block_size=4096):
  """``items``: (key bytes, value bytes) in strictly increasing key order -> an uncompressed table file."""
  out = bytearray()

  def emit(block):
    off = len(out)
    out.extend(block)
    out.append(0)                                          # kNoCompression
    out.extend(struct.pack('<I', mask_crc(crc32c(block + b'\x00'))))
    return off, len(block)
  index = _BlockBuilder(restart_interval=1)
  cur, last_key, prev = _BlockBuilder(), None, None
  for key, value in items:
    if prev is not None and key <= prev:
      raise ValueError('table keys must be strictly increasing: %r after %r' % (key, prev))
    prev = key
    if cur.buf and cur.size() + len(key) + len(value) > block_size:
      off, size = emit(cur.finish())
      index.add(last_key, _put_varint(off) + _put_varint(size))      # the last key of a block is a valid separator
      cur = _BlockBuilder()
    cur.add(key, value)
    last_key = key
  if cur.buf or last_key is None:
    off, size = emit(cur.finish())
    index.add(last_key if last_key is not None else b'', _put_varint(off) + _put_varint(size))
  moff, msize = emit(_BlockBuilder().finish())              # empty metaindex block
  ioff, isize = emit(index.finish())
  footer = _put_varint(moff) + _put_varint(msize) + _put_varint(ioff) + _put_varint(isize)
  footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', TABLE_MAGIC)
  out.extend(footer)
  with open(path, 'wb') as fh:
    fh.write(out)


# ------------------------------------------------------------------------------------------------ tensor bundle
def _data_path(prefix, shard, num_shards):
  return '%s.data-%05d-of-%05d' % (prefix, shard, num_shards)


def list_variables(prefix):
  """[(name, shape, numpy dtype)] of a V2 checkpoint, like tf.train.list_variables."""
  out = []
  for key, value in read_table(prefix + '.index'):
    if key == b'':
      continue
    e = _parse_entry(value)
    out.append((key.decode(), e['shape'], _NP_OF_DT.get(e['dtype'])))
  return out


def read_checkpoint(prefix, names=None, verify=True):
  """{variable name: numpy array} of the V2 checkpoint ``prefix`` (``names``: only these)."""
  items = read_table(prefix + '.index', verify=verify)
  if not items or items[0][0] != b'':
    raise ValueError('%s.index has no bundle header' % prefix)
  header = _parse_message(items[0][1])
  num_shards = header.get(1, [1])[0]
  if header.get(2, [0])[0] != 0:
    raise NotImplementedError('big-endian bundle')
  shards = {}
  out = {}
  for key, value in items[1:]:
    name = key.decode()
    if names is not None and name not in names:
      continue
    e = _parse_entry(value)
    if e['dtype'] not in _NP_OF_DT:
      raise NotImplementedError('variable %s: dtype enum %d is not supported' % (name, e['dtype']))
    if e['shard_id'] not in shards:
      shards[e['shard_id']] = np.memmap(_data_path(prefix, e['shard_id'], num_shards), dtype=np.uint8, mode='r')
    raw = shards[e['shard_id']][e['offset']:e['offset'] + e['size']]
    dt = _NP_OF_DT[e['dtype']]
    count = int(np.prod(e['shape'], dtype=np.int64)) if e['shape'] else 1
    if raw.size != count * dt.itemsize:
      raise ValueError('variable %s: %d bytes on disk, shape %s needs %d' % (name, raw.size, e['shape'], count * dt.itemsize))
    if verify and e['crc32c'] is not None and mask_crc(crc32c(raw)) != e['crc32c']:
      raise ValueError('variable %s: checksum mismatch' % name)
    out[name] = np.frombuffer(raw.tobytes(), dtype=dt).reshape(e['shape']).copy()
  return out


def write_checkpoint(prefix, tensors):
  """Writes {name: array} as the one-shard V2 checkpoint ``prefix`` (.index + .data-00000-of-00001)."""
  os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
  entries = []
  offset = 0
  with open(_data_path(prefix, 0, 1), 'wb') as fh:
    for name in sorted(tensors, key=lambda s: s.encode()):
      a = np.asarray(tensors[name])
      a = a if a.flags.c_contiguous else a.copy(order='C')      # (np.ascontiguousarray would turn a scalar into shape (1,))
      if a.dtype not in _DT_OF_NP:
        raise TypeError('variable %s: dtype %s has no TensorFlow DataType here' % (name, a.dtype))
      raw = a.astype(a.dtype.newbyteorder('<'), copy=False).tobytes()
      fh.write(raw)
      entries.append((name.encode(), _entry_proto(_DT_OF_NP[a.dtype], a.shape, 0, offset, len(raw), mask_crc(crc32c(raw)))))
      offset += len(raw)
  version = _field(1, 0, _put_varint(1))                                      # VersionDef.producer = kTensorBundleVersion
  header = _field(1, 0, _put_varint(1)) + _field(3, 2, _put_varint(len(version)) + version)      # num_shards = 1, LITTLE endian
  write_table(prefix + '.index', [(b'', header)] + entries)


# ------------------------------------------------------------------------------------------------ Saver conventions
def latest_checkpoint(directory):
  """tf.train.latest_checkpoint: the ``model_checkpoint_path`` line of <directory>/checkpoint (relative paths are
  relative to the directory)."""
  state = os.path.join(directory, 'checkpoint')
  if not os.path.isfile(state):
    return None
  with open(state) as fh:
    for line in fh:
      if line.startswith('model_checkpoint_path:'):
        p = line.split(':', 1)[1].strip().strip('"')
        p = p if os.path.isabs(p) else os.path.join(directory, p)
        return p if os.path.isfile(p + '.index') else None
  return None


def _all_checkpoint_paths(directory):
  """the ``all_model_checkpoint_paths`` lines of <directory>/checkpoint, oldest first (names as written)."""
  state = os.path.join(directory, 'checkpoint')
  if not os.path.isfile(state):
    return []
  with open(state) as fh:
    return [l.split(':', 1)[1].strip().strip('"') for l in fh if l.startswith('all_model_checkpoint_paths:')]


RNG_DRAWS_KEY = 'twingan_amd/gp_alpha_draws'      # the one tensor of a checkpoint that is not a variable of the reference


def save(trainer, train_dir, global_step=None, max_to_keep=5):
  """What the reference's Saver leaves for a stage: every model variable (TF names, TF layouts), the non-trainable
  state (moving / renorm statistics, spectral-norm u), ``global_step``, and the shared Adam optimiser's slots
  (``<var>/Adam``, ``<var>/Adam_1``, ``beta1_power``, ``beta2_power``) -> <train_dir>/model.ckpt-<step> + checkpoint
  (the ``max_to_keep`` most recent ones are retained)."""
  store = trainer.store
  step = int(trainer.global_step if global_step is None else global_step)
  tensors = {k: v.detach().float().cpu().numpy() for k, v in store.state_dict(include_state=True).items()}
  for k, (m, v) in store.adam_dict().items():
    tensors[k + '/Adam'] = m.detach().float().cpu().numpy()
    tensors[k + '/Adam_1'] = v.detach().float().cpu().numpy()
  t = int(trainer.adam_t)
  tensors['beta1_power'] = np.float32(trainer.cfg.adam_beta1 ** (t + 1))       # TF keeps beta^(t+1) after t applies
  tensors['beta2_power'] = np.float32(trainer.cfg.adam_beta2 ** (t + 1))
  tensors['global_step'] = np.int64(step)
  # image_generation.py:622-623: a global (hence saved) int32 variable; every session.run adds 1 to it and applies the
  # shared Adam once, so it is also the number of Adam applies -- which the beta powers stop encoding once they
  # underflow (float32 0.5^(t+1) is exactly 0 from t = 149)
  tensors['n_critic_counter'] = np.int32(trainer.n_critic_counter)
  # not a variable of the reference (its tf.random_uniform draws are stateful in the session, never saved): the draw counter
  # of this clone's device generator (the gradient penalty's interpolation weights, ops.uniform), so that a resumed run
  # continues the sequence instead of replaying it from the first draw.  A TensorFlow-written checkpoint simply lacks it.
  if getattr(trainer, '_rng_state', None) is not None:
    tensors[RNG_DRAWS_KEY] = np.int64(int(trainer._rng_state[0].item()))
  name = 'model.ckpt-%d' % step
  kept = [n for n in _all_checkpoint_paths(train_dir) if n != name]
  write_checkpoint(os.path.join(train_dir, name), tensors)
  kept.append(name)
  # tf.train.Saver(max_to_keep=5), the default slim.learning.train builds (model_inheritor.py:1119-1130): the state file
  # lists the retained checkpoints oldest first, older ones are deleted
  while max_to_keep and len(kept) > max_to_keep:
    old = kept.pop(0)
    base = old if os.path.isabs(old) else os.path.join(train_dir, old)
    for suffix in ('.index', '.data-00000-of-00001'):
      if os.path.isfile(base + suffix):
        os.remove(base + suffix)
  with open(os.path.join(train_dir, 'checkpoint'), 'w') as fh:
    fh.write('model_checkpoint_path: "%s"\n' % name)
    for n in kept:
      fh.write('all_model_checkpoint_paths: "%s"\n' % n)
  return os.path.join(train_dir, name)


def init_from_checkpoint(trainer, checkpoint_path, checkpoint_exclude_scopes=None, ignore_missing_vars=False,
                         train_dir=None):
  """model/model_inheritor.py:576-644 (_get_init_fn + slim.assign_from_checkpoint_fn): restore the MODEL variables
  (slim.get_model_variables(): not the optimiser slots, not global_step, and not the attention gate ``sa_gamma``, the
  one plain tf.get_variable of the path -- params.is_model_variable) whose names do not start with an excluded
  scope from ``checkpoint_path``
  (a checkpoint prefix, or a directory -> its latest checkpoint).  Nothing is restored when ``train_dir`` already holds
  a checkpoint (the run resumes from that one instead).  A variable the checkpoint lacks is an error unless
  ``ignore_missing_vars`` (growing stages: the new resolution's layers keep their fresh initialisation,
  pggan_runner.py:136-146); a shape mismatch is always an error, as in TensorFlow.  Returns the restored names."""
  if checkpoint_path is None:
    return []
  if train_dir is not None and latest_checkpoint(train_dir):
    return []
  prefix = latest_checkpoint(checkpoint_path) if os.path.isdir(checkpoint_path) else checkpoint_path
  if prefix is None:
    raise FileNotFoundError('no checkpoint in %s' % checkpoint_path)
  exclusions = [s.strip() for s in (checkpoint_exclude_scopes or '').split(',') if s.strip()]
  from .params import is_model_variable
  store = trainer.store
  wanted = [k for k in list(store.specs) + list(store.state_specs)
            if is_model_variable(k) and not any(k.startswith(e) for e in exclusions)]
  available = {name: shape for name, shape, _ in list_variables(prefix)}
  missing = [k for k in wanted if k not in available]
  if missing and not ignore_missing_vars:
    raise KeyError('checkpoint %s lacks %d variable(s), e.g. %s' % (prefix, len(missing), missing[0]))
  take = [k for k in wanted if k in available]
  arrays = read_checkpoint(prefix, names=set(take))
  import torch
  sd = {}
  for k in take:
    want = tuple(store.specs[k]['shape']) if k in store.specs else store.state_shape(k)
    if tuple(arrays[k].shape) != want and not (k in store.state and tuple(arrays[k].shape) == tuple(store.state[k].shape)):
      raise ValueError('variable %s: checkpoint shape %s, model shape %s' % (k, arrays[k].shape, want))
    sd[k] = torch.from_numpy(arrays[k].astype(np.float32))
  store.load_state_dict(sd, strict=False)
  return sorted(take)


def restore(trainer, prefix):
  """tf.train.Saver.restore of a checkpoint this trainer's stage wrote (resuming a run, model_inheritor.py:596-602:
  a checkpoint in train_dir takes precedence over --checkpoint_path): model variables, state, the shared optimiser's
  slots and beta powers, global_step.  Every variable of the model must be present with its shape."""
  import math
  import torch
  store = trainer.store
  arrays = read_checkpoint(prefix)
  names = list(store.specs) + list(store.state_specs)
  missing = [k for k in names if k not in arrays]
  if missing:
    raise KeyError('checkpoint %s lacks %d variable(s), e.g. %s' % (prefix, len(missing), missing[0]))
  store.load_state_dict({k: torch.from_numpy(arrays[k].astype(np.float32)) for k in names}, strict=False)
  slots = {k: (arrays[k + '/Adam'], arrays[k + '/Adam_1']) for k in store.specs if k + '/Adam' in arrays}
  store.load_adam_dict(slots)
  if 'global_step' in arrays:
    trainer.global_step = int(arrays['global_step'])
    trainer.n_critic_counter = trainer.global_step * trainer.cfg.n_critic
  if 'n_critic_counter' in arrays:
    trainer.n_critic_counter = int(arrays['n_critic_counter'])
  trainer.set_adam_step(_adam_applies(arrays, trainer.cfg, trainer.n_critic_counter))
  if RNG_DRAWS_KEY in arrays and getattr(trainer, '_rng_state', None) is not None:
    trainer._rng_state[0] = int(arrays[RNG_DRAWS_KEY])      # continue the device generator's sequence (see save)
  return trainer.global_step


def _adam_applies(arrays, cfg, counter):
  """Number of applies t the shared Adam optimiser has made.  The reference applies the one optimiser exactly once per
  n_critic_counter increment (image_generation.py:640-652), so the counter IS t when the checkpoint has it.  Without it:
  TF keeps beta^(t+1) in float32 (repeated float32 products, off by ~1.3e-8 relative per apply), so invert whichever power
  is still a normal number (beta1 = 0.5 underflows to exactly 0 at t = 149, beta2 = 0.999 near t = 87 000) -- exact for
  short runs, within a step or two for long ones; after that the caller's counter (global_step * n_critic)."""
  import math
  if 'n_critic_counter' in arrays:      # exact; TF's float32 beta powers drift by ~1.3e-5 steps per apply (half a step at 40 k)
    return max(0, int(arrays['n_critic_counter']))
  for key, beta in (('beta2_power', cfg.adam_beta2), ('beta1_power', cfg.adam_beta1)):
    if key in arrays and 0.0 < beta < 1.0:
      p = float(arrays[key])
      if 1e-30 < p <= 1.0:
        return max(0, int(round(math.log(p) / math.log(beta))) - 1)
  return max(0, int(counter))
