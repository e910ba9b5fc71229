"""twingan_amd/checkpoint.py on CPU: the TensorFlow V2 checkpoint (tensor bundle) reader / writer and the reference's
warm-start rule (model/model_inheritor.py:576-644).  No TensorFlow exists here, so the format is held to its published
test vectors (CRC-32C: RFC 3720 B.4; LevelDB's masked-CRC and varint conventions), to a table assembled BY HAND in this
file (independent of the writer), and to round trips."""
import os
import struct

import numpy as np
import pytest
import torch

from twingan_amd import checkpoint as C


def test_crc32c_known_vectors():
  assert C.crc32c(b'123456789') == 0xe3069283
  assert C.crc32c(b'\x00' * 32) == 0x8a9136aa          # RFC 3720 B.4
  assert C.crc32c(b'\xff' * 32) == 0x62a8ab43
  assert C.crc32c(bytes(range(32))) == 0x46dd794e
  assert C.crc32c(b'') == 0
  # leveldb crc32c_test: Mask is not an involution and is undone by Unmask
  crc = C.crc32c(b'foo')
  assert C.mask_crc(crc) != crc and C.mask_crc(C.mask_crc(crc)) != crc
  assert C.unmask_crc(C.mask_crc(crc)) == crc and C.unmask_crc(C.unmask_crc(C.mask_crc(C.mask_crc(crc)))) == crc


def test_crc32c_lane_parallel_path_equals_serial():
  rng = np.random.RandomState(0)
  for n in (64 * 1024, 64 * 1024 + 1, 300001, 1 << 20):
    data = rng.randint(0, 256, n, dtype=np.uint8).tobytes()
    serial = C._crc_bytes(0xffffffff, data) ^ 0xffffffff
    assert C.crc32c(data) == serial, n


def test_varint_and_snappy():
  for v in (0, 1, 127, 128, 300, 2 ** 32 - 1, 2 ** 63):
    enc = C._put_varint(v)
    assert C._get_varint(enc, 0) == (v, len(enc))
  assert C._put_varint(300) == b'\xac\x02'
  # snappy: length 11, literal "abc", copy (offset 3, length 8) -> "abcabcabcab"
  comp = bytes([11, (3 - 1) << 2]) + b'abc' + bytes([((8 - 4) << 2) | 1, 3])
  assert C._snappy_uncompress(comp) == b'abcabcabcab'


def _hand_table(path, entries):
  """A table file assembled without twingan_amd.checkpoint's writer: one data block, no key sharing, every entry a
  restart point; index block with one entry."""
  def block(kvs):
    body, restarts = b'', []
    for k, v in kvs:
      restarts.append(len(body))
      body += bytes([0, len(k), len(v)]) + k + v          # all lengths < 128 here: one-byte varints
    return body + b''.join(struct.pack('<I', r) for r in restarts) + struct.pack('<I', len(restarts))

  def trailer(b):
    return b + b'\x00' + struct.pack('<I', C.mask_crc(C.crc32c(b + b'\x00')))
  out = b''
  data = block(entries)
  d_off, d_size = len(out), len(data)
  out += trailer(data)
  meta = block([])
  m_off, m_size = len(out), len(meta)
  out += trailer(meta)
  index = block([(entries[-1][0], bytes([d_off, d_size]))])
  i_off, i_size = len(out), len(index)
  out += trailer(index)
  footer = bytes([m_off, m_size]) + C._put_varint(i_off) + C._put_varint(i_size)
  out += footer + b'\x00' * (40 - len(footer)) + struct.pack('<Q', 0xdb4775248b80fb57)
  with open(path, 'wb') as fh:
    fh.write(out)


def test_reader_on_a_hand_assembled_bundle(tmp_path):
  """A two-variable bundle written byte by byte from the format description (protobuf fields spelt out)."""
  w = np.arange(6, dtype='<f4').reshape(2, 3)
  step = np.array(7, dtype='<i8')
  raw_w, raw_s = w.tobytes(), step.tobytes()
  with open(tmp_path / 'm.ckpt.data-00000-of-00001', 'wb') as fh:
    fh.write(raw_w + raw_s)
  header = b'\x08\x01' + b'\x1a\x02\x08\x01'                                   # num_shards = 1; version { producer: 1 }
  e_w = (b'\x08\x01' +                                                        # dtype = DT_FLOAT
         b'\x12\x08' + b'\x12\x02\x08\x02' + b'\x12\x02\x08\x03' +            # shape { dim {size: 2} dim {size: 3} }
         b'\x28' + bytes([len(raw_w)]) +                                      # size (offset 0 and shard 0 are defaults)
         b'\x35' + struct.pack('<I', C.mask_crc(C.crc32c(raw_w))))            # crc32c (fixed32)
  e_s = (b'\x08\x09' + b'\x12\x00' + b'\x20' + bytes([len(raw_w)]) + b'\x28\x08' +
         b'\x35' + struct.pack('<I', C.mask_crc(C.crc32c(raw_s))))            # DT_INT64 scalar at offset 24
  _hand_table(str(tmp_path / 'm.ckpt.index'), [(b'', header), (b'global_step', e_s), (b'scope/weights', e_w)])
  got = C.read_checkpoint(str(tmp_path / 'm.ckpt'))
  assert set(got) == {'global_step', 'scope/weights'}
  assert got['global_step'].shape == () and int(got['global_step']) == 7
  np.testing.assert_array_equal(got['scope/weights'], w)
  assert C.list_variables(str(tmp_path / 'm.ckpt')) == [('global_step', (), np.dtype('<i8')), ('scope/weights', (2, 3), np.dtype('<f4'))]
  # a flipped data byte is caught by the entry's checksum
  with open(tmp_path / 'm.ckpt.data-00000-of-00001', 'r+b') as fh:
    fh.seek(3)
    fh.write(b'\x7f')
  with pytest.raises(ValueError, match='checksum'):
    C.read_checkpoint(str(tmp_path / 'm.ckpt'))


def test_writer_reader_round_trip_many_blocks(tmp_path):
  rng = np.random.RandomState(1)
  tensors = {'net/block_%dx%d/conv_%d/weights' % (h, h, i): rng.randn(3, 3, 4, 5).astype(np.float32)
             for h in (4, 8, 16, 32, 64) for i in range(40)}      # > 4 KB of index: several data blocks, shared key prefixes
  tensors.update({'global_step': np.int64(123456789012), 'beta1_power': np.float32(0.25), 'flag': np.array([True, False]),
                  'half': rng.randn(7).astype(np.float16), 'big': rng.randn(300, 301).astype(np.float32)})
  prefix = str(tmp_path / 'model.ckpt-5')
  C.write_checkpoint(prefix, tensors)
  assert len(C.read_table(prefix + '.index')) == len(tensors) + 1
  got = C.read_checkpoint(prefix)
  assert set(got) == set(tensors)
  for k, v in tensors.items():
    np.testing.assert_array_equal(got[k], np.asarray(v))
    assert got[k].dtype == np.asarray(v).dtype and got[k].shape == np.asarray(v).shape
  only = C.read_checkpoint(prefix, names={'big'})
  assert list(only) == ['big']
  # index corruption is caught by the block checksum
  with open(prefix + '.index', 'r+b') as fh:
    fh.seek(10)
    b = fh.read(1)
    fh.seek(10)
    fh.write(bytes([b[0] ^ 1]))
  with pytest.raises(ValueError):
    C.read_checkpoint(prefix)


def test_save_and_warm_start_follow_the_reference_rules(tmp_path):
  """Saver naming (model.ckpt-<step>, ``checkpoint`` state file, Adam slots) and _get_init_fn's rule: model variables
  only, excluded scopes, ignore_missing_vars for a growing stage, nothing restored when train_dir already has a
  checkpoint (model/model_inheritor.py:596-602,604-622,641-644)."""
  from twingan_amd import Config
  from twingan_amd.twingan import Trainer
  small = Trainer(Config(hw=4, max_ch=8, precision='fp32'), device='cpu', seed=1)
  small.global_step, small.adam_t = 11, 22
  with torch.no_grad():
    small.store.m['g'].fill_(0.5)
    small.store.v['d'].fill_(0.25)
  d4 = str(tmp_path / '4')
  path = C.save(small, d4)
  assert path.endswith('model.ckpt-11') and C.latest_checkpoint(d4) == path
  names = {n for n, _, _ in C.list_variables(path)}
  sd = small.store.state_dict(include_state=True)
  assert set(sd) <= names and 'global_step' in names and 'beta1_power' in names
  some = next(iter(small.store.specs))
  assert some + '/Adam' in names and some + '/Adam_1' in names
  back = C.read_checkpoint(path)
  assert back['global_step'].shape == () and int(back['global_step']) == 11
  assert abs(float(back['beta1_power']) - small.cfg.adam_beta1 ** 23) < 1e-7
  for k, v in sd.items():
    np.testing.assert_array_equal(back[k], v.numpy())

  grown = Trainer(Config(hw=8, max_ch=8, precision='fp32', is_growing=True), device='cpu', seed=2)
  fresh = grown.store.state_dict(include_state=True)
  with pytest.raises(KeyError):
    C.init_from_checkpoint(grown, d4)                                  # the 8x8 layers are not in the 4x4 checkpoint
  loaded = C.init_from_checkpoint(grown, d4, ignore_missing_vars=True, train_dir=str(tmp_path / '4to8'))
  after = grown.store.state_dict(include_state=True)
  assert loaded and set(loaded) == {k for k in fresh if k in sd}
  for k in fresh:
    if k in loaded:
      assert torch.equal(after[k], sd[k])
    else:
      assert torch.equal(after[k], fresh[k])
  assert float(grown.store.m['g'].abs().max()) == 0                    # optimiser slots are not model variables
  # excluded scopes keep their fresh values
  again = Trainer(Config(hw=8, max_ch=8, precision='fp32', is_growing=True), device='cpu', seed=2)
  got = C.init_from_checkpoint(again, path, checkpoint_exclude_scopes='discriminator_s, generator', ignore_missing_vars=True)
  assert got and not any(k.startswith(('discriminator_s', 'generator')) for k in got)
  # a checkpoint in train_dir wins: nothing is restored from checkpoint_path
  C.save(again, str(tmp_path / '4to8'))
  third = Trainer(Config(hw=8, max_ch=8, precision='fp32', is_growing=True), device='cpu', seed=3)
  assert C.init_from_checkpoint(third, d4, ignore_missing_vars=True, train_dir=str(tmp_path / '4to8')) == []
  for tr in (small, grown, again, third):
    tr.close()


def test_run_progressive_directory_protocol(tmp_path):
  """pggan_runner.py:100-160 on checkpoint files: one directory per stage, warm start from the previous stage's
  directory (ignore_missing_vars = is_growing), skip of stages that are already trained.  (No training steps here:
  the CPU has no kernels; the GPU twin is test_progressive_stages_with_warm_start_match_oracle.)"""
  from twingan_amd import Config
  from twingan_amd.runner import run_progressive
  base = Config(hw=4, max_ch=8, precision='fp32')
  table = {4: 2, 8: 2}
  root = str(tmp_path / 'run')
  ends = {}
  state, hist = run_progressive(base, None, 4, 8, table, num_images_per_resolution=4, device='cpu', seed=7, max_steps_per_stage=0,
                                train_dir=root, on_stage_end=lambda name, tr: ends.__setitem__(name, tr.store.state_dict()))
  assert [h['stage'] for h in hist] == ['4', '4to8', '8'] and sorted(os.listdir(root)) == ['4', '4to8', '8']
  assert hist[1]['warm_started'] > 0 and hist[2]['warm_started'] == len(ends['8'])      # the stable stage finds every variable
  for k, v in ends['8'].items():                                                          # ... with the growing stage's values
    assert torch.equal(v, ends['4to8'][k])
  shared = [k for k in ends['4'] if k in ends['4to8'] and ends['4'][k].shape == ends['4to8'][k].shape]
  assert shared and all(torch.equal(ends['4'][k], ends['4to8'][k]) for k in shared)
  on_disk = C.read_checkpoint(C.latest_checkpoint(os.path.join(root, '8')))
  for k, v in ends['8'].items():
    np.testing.assert_array_equal(on_disk[k], v.numpy())
  # a second invocation finds every stage trained and skips it
  _, again = run_progressive(base, None, 4, 8, table, num_images_per_resolution=4, device='cpu', seed=8, max_steps_per_stage=0,
                             train_dir=root)
  assert all(h.get('skipped') for h in again)


def test_warm_start_restores_model_variables_only(tmp_path):
  """slim.get_model_variables() (model_inheritor.py:612-614) holds everything the layers create -- including the
  spectral-norm ``u`` (libs/sn.py:56 under the layer scope's model-variable getter, pinned live by
  test_reference_live.py::test_warm_start_set_is_slims_model_variables) -- but not ``sa_gamma``
  (libs/self_attention.py:68, plain tf.get_variable): a stage's warm start carries ``u`` over and leaves ``sa_gamma`` at
  its fresh initialisation (both the in-memory and the file-based warm start); a full restore of a run
  (checkpoint.restore) loads both."""
  from twingan_amd import Config
  from twingan_amd.runner import warm_start
  from twingan_amd.twingan import Trainer
  cfg = Config(hw=8, max_ch=8, precision='fp32', spectral_norm=True, do_self_attention=True, self_attention_hw=8)
  a = Trainer(cfg, device='cpu', seed=1)
  with torch.no_grad():
    for k, p in a.store.P.items():
      if k.endswith('/sa_gamma'):
        p.fill_(0.7)
  sd = a.store.state_dict(include_state=True)
  us = [k for k in sd if k.endswith('/u')]
  gates = [k for k in sd if k.endswith('/sa_gamma')]
  assert us and gates
  C.save(a, str(tmp_path / 's'))
  for how in ('memory', 'files', 'restore'):
    b = Trainer(cfg, device='cpu', seed=2)
    fresh = b.store.state_dict(include_state=True)
    assert not any(torch.equal(fresh[k], sd[k]) for k in us)          # a different seed draws a different u
    if how == 'memory':
      loaded = warm_start(b, sd)
    elif how == 'files':
      loaded = C.init_from_checkpoint(b, str(tmp_path / 's'))
    else:
      C.restore(b, C.latest_checkpoint(str(tmp_path / 's')))
      loaded = list(sd)
    after = b.store.state_dict(include_state=True)
    for k in sd:
      if k in gates and how != 'restore':
        assert k not in loaded and torch.equal(after[k], fresh[k]), (how, k)
      else:
        assert k in loaded and torch.equal(after[k], sd[k]), (how, k)
    b.close()
  a.close()


@pytest.mark.parametrize('applies', [3, 148, 149, 200, 5000, 200000])
def test_restore_recovers_the_adam_step_after_the_beta_powers_underflow(tmp_path, applies):
  """TF keeps beta^(t+1) in float32: beta1 = 0.5 is exactly 0 from t = 149 (75 G+D steps), beta2 = 0.999 fades near
  t = 87 000.  restore() inverts whichever power is still a normal number and otherwise reads the n_critic counter the
  reference saves next to them (image_generation.py:622-623; one Adam apply per increment)."""
  from twingan_amd import Config
  from twingan_amd.twingan import Trainer
  a = Trainer(Config(hw=4, max_ch=8, precision='fp32'), device='cpu', seed=1)
  a.set_adam_step(applies)
  a.n_critic_counter, a.global_step = applies, applies // a.cfg.n_critic
  path = C.save(a, str(tmp_path / 'run'))
  back = C.read_checkpoint(path)
  assert int(back['n_critic_counter']) == applies and back['n_critic_counter'].dtype == np.int32
  if applies >= 149:
    assert float(back['beta1_power']) == 0.0
  b = Trainer(Config(hw=4, max_ch=8, precision='fp32'), device='cpu', seed=2)
  C.restore(b, path)
  assert (b.adam_t, b.n_critic_counter, b.global_step) == (applies, applies, applies // a.cfg.n_critic)
  # a checkpoint without the counter (hand-made / older): the powers while they last, then global_step * n_critic
  arrays = {k: v for k, v in back.items() if k != 'n_critic_counter'}
  assert C._adam_applies(arrays, a.cfg, (applies // a.cfg.n_critic) * a.cfg.n_critic) in (applies, applies - applies % a.cfg.n_critic)
  a.close(); b.close()


def test_renorm_weights_are_scalars_in_checkpoints(tmp_path):
  """libs/batch_norm.py:237,246 (and tf.layers.BatchNormalization) create renorm_mean_weight / renorm_stddev_weight with
  shape (): state_dict and the checkpoint files carry them as scalars, restore / init_from_checkpoint take a scalar (what a
  reference-written file holds) and still accept this repo's older one-element form."""
  from twingan_amd import Config
  from twingan_amd.twingan import Trainer
  for nt in ('batch_renorm', 'batch_renorm_native'):
    cfg = Config(hw=8, max_ch=8, precision='fp32', generator_norm_type=nt)
    a = Trainer(cfg, device='cpu', seed=1)
    ws = [k for k in a.store.state_specs if k.rsplit('/', 1)[-1].startswith(('renorm_mean_weight', 'renorm_stddev_weight'))]
    assert ws and set(ws) == a.store.scalar_state
    for i, k in enumerate(ws):
      a.store.state[k].fill_(0.25 + 0.01 * i)
    sd = a.store.state_dict(include_state=True)
    assert all(tuple(sd[k].shape) == () for k in ws)
    path = C.save(a, str(tmp_path / nt))
    back = C.read_checkpoint(path)
    assert all(back[k].shape == () for k in ws)
    for how in ('restore', 'init', 'legacy'):
      b = Trainer(cfg, device='cpu', seed=2)
      if how == 'restore':
        C.restore(b, path)
      elif how == 'init':
        C.init_from_checkpoint(b, str(tmp_path / nt))
      else:      # the one-element form older files of this repo hold
        b.store.load_state_dict({k: sd[k].reshape(1) for k in ws}, strict=False)
      for i, k in enumerate(ws):
        assert tuple(b.store.state[k].shape) == (1,) and abs(float(b.store.state[k]) - (0.25 + 0.01 * i)) < 1e-7, (how, k)
      b.close()
    a.close()


def test_saver_keeps_the_most_recent_checkpoints(tmp_path):
  """tf.train.Saver(max_to_keep=5) as slim.learning.train builds it: the state file lists the retained checkpoints
  oldest first, older files are deleted, latest_checkpoint follows ``model_checkpoint_path``."""
  from twingan_amd import Config
  from twingan_amd.twingan import Trainer
  tr = Trainer(Config(hw=4, max_ch=8, precision='fp32'), device='cpu', seed=1)
  d = str(tmp_path / 'run')
  for step in (10, 20, 30, 40):
    C.save(tr, d, global_step=step, max_to_keep=3)
  assert C._all_checkpoint_paths(d) == ['model.ckpt-20', 'model.ckpt-30', 'model.ckpt-40']
  assert C.latest_checkpoint(d).endswith('model.ckpt-40')
  assert not os.path.exists(os.path.join(d, 'model.ckpt-10.index')) and os.path.exists(os.path.join(d, 'model.ckpt-20.index'))
  C.save(tr, d, global_step=40, max_to_keep=3)                          # re-saving a step does not duplicate its entry
  assert C._all_checkpoint_paths(d) == ['model.ckpt-20', 'model.ckpt-30', 'model.ckpt-40']
  tr.close()


def test_resume_continues_the_gradient_penalty_draws(tmp_path):
  """The gradient penalty's interpolation weights come from the clone's device generator (Philox keyed by (seed, draw counter),
  ops.uniform); the counter travels in the checkpoint (the one tensor there that is no variable of the reference: its
  tf.random_uniform state lives in the session), so a resumed run continues the sequence instead of replaying it from the
  first draw; a checkpoint without it (one TensorFlow wrote) leaves the counter alone."""
  from twingan_amd import Config
  from twingan_amd.twingan import Trainer
  a = Trainer(Config(hw=4, max_ch=8, precision='fp32'), device='cpu', seed=1)
  a._rng_state[0] = 12345
  d = str(tmp_path / 'run')
  path = C.save(a, d, global_step=7)
  arrays = C.read_checkpoint(path)
  assert int(arrays[C.RNG_DRAWS_KEY]) == 12345
  b = Trainer(Config(hw=4, max_ch=8, precision='fp32'), device='cpu', seed=2)
  assert int(b._rng_state[0]) == 0
  C.restore(b, path)
  assert int(b._rng_state[0]) == 12345 and b.global_step == 7
  # a bundle without the key: written from the same tensors minus it
  del arrays[C.RNG_DRAWS_KEY]
  C.write_checkpoint(os.path.join(d, 'plain.ckpt-7'), arrays)
  c = Trainer(Config(hw=4, max_ch=8, precision='fp32'), device='cpu', seed=3)
  c._rng_state[0] = 5
  C.restore(c, os.path.join(d, 'plain.ckpt-7'))
  assert int(c._rng_state[0]) == 5
  for t in (a, b, c):
    t.close()
