"""twingan_amd/data.py on CPU (host logic; the preprocessing kernel itself is in tests/test_gpu_data.py) and the
NumPy restatement of the trainer's image preprocessing against the fixture computed by the reference's own
preprocess_image (tests/golden/preprocess_hw32.npz, tools/make_golden.py --preprocess)."""
import io
import os
import struct
import sys

import numpy as np
import pytest
import torch

from oracle import np_ops as N
from twingan_amd import data as D

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'preprocess_hw32.npz')


def _cases():
  g = np.load(GOLD)
  i = 0
  while 'img%d' % i in g:
    yield i, g['img%d' % i], str(g['mode%d' % i]), g['par%d' % i], g['out%d' % i], int(g['hw'])
    i += 1


def test_preprocess_oracle_hits_the_reference_fixture():
  """oracle.np_ops.preprocess_image == preprocessing/danbooru_preprocessing.preprocess_image as executed when the fixture
  was made (10 images: PAD / CROP / RESHAPE, portrait / landscape / square, both colour orders, flips, evaluation)."""
  seen = set()
  for i, img, mode, par, want, hw in _cases():
    flip, sat_first, delta, factor, training, sel = par
    got = N.preprocess_image(img, hw, mode, bool(training), flip=bool(flip), saturation_first=bool(sat_first), delta=delta,
                             factor=factor)
    assert np.abs(got - want).max() < 1e-12, i
    assert got.min() >= 0 and got.max() <= 1
    seen.add((mode, bool(flip), bool(sat_first), bool(training)))
  assert {m for m, *_ in seen} == {'PAD', 'CROP', 'RESHAPE'} and len(seen) >= 6


@pytest.mark.skipif(not os.path.isdir('/root/reference'), reason='the reference tree is only mounted in the build container')
def test_preprocess_oracle_against_the_live_reference():
  from oracle import ref_runner
  rng = np.random.RandomState(5)
  for k, (h, w, mode) in enumerate([(45, 61, 'PAD'), (80, 33, 'CROP'), (25, 25, 'PAD'), (19, 70, 'RESHAPE')]):
    img = rng.randint(0, 256, (h, w, 3), dtype=np.uint8)
    want, dr = ref_runner.run_preprocess(img, 24, mode, True, seed=100 + k)
    ap = dict(dr['applied'])
    got = N.preprocess_image(img, 24, mode, True, flip=dr['flip_uniform'] < 0.5,
                             saturation_first=dr['applied'][0][0] == 'saturation', delta=ap['brightness'],
                             factor=ap['saturation'])
    assert np.abs(got - want).max() < 1e-12
    # apply_with_random_selector: ordering 0 is the only one with brightness first in fast mode
    assert (dr['sel'] == 0) == (dr['applied'][0][0] == 'brightness')


def test_hsv_round_trip_matches_colorsys():
  import colorsys
  rgb = np.random.RandomState(0).rand(500, 3)
  hsv = N.rgb_to_hsv(rgb)
  ref = np.array([colorsys.rgb_to_hsv(*p) for p in rgb])
  assert np.abs(hsv - ref).max() < 1e-12
  assert np.abs(N.hsv_to_rgb(ref) - np.array([colorsys.hsv_to_rgb(*p) for p in ref])).max() < 1e-12


def test_tfrecord_round_trip_and_corruption(tmp_path):
  payloads = [b'', b'a', os.urandom(1000), b'x' * 70000]
  path = str(tmp_path / 'train-00000-of-00001')
  D.write_tfrecords(path, payloads)
  assert list(D.read_tfrecords(path, verify=True)) == payloads
  raw = open(path, 'rb').read()
  # layout of the first (empty) record: length 0, crc of the length, crc of the empty payload
  assert raw[:8] == struct.pack('<Q', 0) and len(raw) == sum(16 + len(p) for p in payloads)
  bad = bytearray(raw)
  bad[40] ^= 1
  open(path, 'wb').write(bad)
  with pytest.raises(ValueError):
    list(D.read_tfrecords(path, verify=True))


def test_example_codec_matches_protobuf_wire_format():
  ex = D.encode_example({'image/encoded': b'\x00\x01jpegbytes', 'image/format': 'jpeg', 'image/height': 17,
                         'labels': [1, 2, -3], 'embedding': [0.5, -2.0]})
  got = D.decode_example(ex)
  assert got['image/encoded'] == [b'\x00\x01jpegbytes'] and got['image/format'] == [b'jpeg']
  assert got['image/height'] == [17] and got['labels'] == [1, 2, -3] and got['embedding'] == [0.5, -2.0]
  # the same message built with the protobuf runtime from a descriptor spelt out here (independent of the encoder)
  from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
  fd = descriptor_pb2.FileDescriptorProto(name='ex_test.proto', package='t', syntax='proto3')

  def msg(name, fields):
    m = fd.message_type.add(name=name)
    for fname, num, typ, label, tname in fields:
      f = m.field.add(name=fname, number=num, type=typ, label=label)
      if tname:
        f.type_name = tname
    return m
  T = descriptor_pb2.FieldDescriptorProto
  msg('BytesList', [('value', 1, T.TYPE_BYTES, T.LABEL_REPEATED, None)])
  msg('FloatList', [('value', 1, T.TYPE_FLOAT, T.LABEL_REPEATED, None)])
  msg('Int64List', [('value', 1, T.TYPE_INT64, T.LABEL_REPEATED, None)])
  msg('Feature', [('bytes_list', 1, T.TYPE_MESSAGE, T.LABEL_OPTIONAL, '.t.BytesList'),
                  ('float_list', 2, T.TYPE_MESSAGE, T.LABEL_OPTIONAL, '.t.FloatList'),
                  ('int64_list', 3, T.TYPE_MESSAGE, T.LABEL_OPTIONAL, '.t.Int64List')])
  entry = msg('Entry', [('key', 1, T.TYPE_STRING, T.LABEL_OPTIONAL, None), ('value', 2, T.TYPE_MESSAGE, T.LABEL_OPTIONAL, '.t.Feature')])
  msg('Features', [('feature', 1, T.TYPE_MESSAGE, T.LABEL_REPEATED, '.t.Entry')])
  msg('Example', [('features', 1, T.TYPE_MESSAGE, T.LABEL_OPTIONAL, '.t.Features')])
  pool = descriptor_pool.DescriptorPool()
  pool.Add(fd)
  Example = message_factory.GetMessageClass(pool.FindMessageTypeByName('t.Example'))
  m = Example()
  m.ParseFromString(ex)                        # the runtime accepts the bytes ...
  by_key = {e.key: e.value for e in m.features.feature}
  assert list(by_key['labels'].int64_list.value) == [1, 2, -3] and list(by_key['embedding'].float_list.value) == [0.5, -2.0]
  assert by_key['image/encoded'].bytes_list.value[0] == b'\x00\x01jpegbytes'
  assert D.decode_example(m.SerializeToString()) == got      # ... and what it writes decodes to the same features


def test_image_only_dataset_and_decoder(tmp_path):
  from PIL import Image
  rng = np.random.RandomState(3)
  def smooth(h, w):      # JPEG keeps smooth content within a few grey levels
    yy, xx = np.mgrid[0:h, 0:w]
    return np.stack([(4 * yy + 2 * xx) % 256, (3 * xx + 40) % 256, (yy * xx // 4 + 10) % 200], axis=-1).astype(np.uint8)
  imgs = [rng.randint(0, 256, (20 + 3 * i, 30 - i, 3), dtype=np.uint8) if i % 2 else smooth(20 + 3 * i, 30 - i)
          for i in range(5)]
  recs = []
  for i, a in enumerate(imgs):
    buf = io.BytesIO()
    Image.fromarray(a).save(buf, format='PNG' if i % 2 else 'JPEG', quality=95)
    recs.append(D.image_example(buf.getvalue(), 'png' if i % 2 else 'jpeg', 'f%d' % i))
  D.write_tfrecords(str(tmp_path / 'train-00000-of-00002'), recs[:3])
  D.write_tfrecords(str(tmp_path / 'train-00001-of-00002'), recs[3:])
  D.write_tfrecords(str(tmp_path / 'validation-00000-of-00001'), recs[:1])
  ds = D.ImageOnlyDataset(str(tmp_path), 'train')
  assert len(ds.files) == 2
  got = list(ds)
  assert [n for _, n in got] == ['f%d' % i for i in range(5)]
  for i, (a, _) in enumerate(got):
    assert a.dtype == np.uint8 and a.shape == imgs[i].shape
    if i % 2:
      np.testing.assert_array_equal(a, imgs[i])                # PNG is lossless
    else:
      assert np.abs(a.astype(int) - imgs[i].astype(int)).mean() < 12      # JPEG, quality 95, smooth content
  with pytest.raises(FileNotFoundError):
    D.ImageOnlyDataset(str(tmp_path), 'test')


def test_source_rect_and_draws():
  assert D.source_rect(37, 53, 'PAD') == (-8, 0, 53, 53) and D.source_rect(64, 40, 'PAD') == (0, -12, 64, 64)
  assert D.source_rect(37, 53, 'CROP') == (0, 8, 37, 37) and D.source_rect(20, 33, 'RESHAPE') == (0, 0, 20, 33)
  aug = D.draw_augmentation(4000, np.random.default_rng(0))
  assert set(np.unique(aug[:, 0])) == {0.0, 1.0} and abs(aug[:, 0].mean() - 0.5) < 0.05
  assert abs(aug[:, 1].mean() - 0.75) < 0.05                   # 3 of the 4 orderings start with saturation
  assert aug[:, 2].min() >= -32 / 255 and aug[:, 2].max() < 32 / 255 and 0.5 <= aug[:, 3].min() and aug[:, 3].max() < 1.5
  pre = D.Preprocessor(16, device='cpu')
  with pytest.raises(RuntimeError):
    pre([np.zeros((4, 4, 3), np.uint8)])


MODES = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'preprocess_modes_hw32.npz')


def _mode_case(g, i):
  par = g['par%d' % i]
  crop = tuple(int(v) for v in g['crop%d' % i]) if g['crop%d' % i][0] >= 0 else None
  moff = tuple(int(v) for v in g['moff%d' % i]) if g['moff%d' % i][0] >= 0 else None
  return dict(img=g['img%d' % i], want=g['out%d' % i], mode=str(g['mode%d' % i]), cs=str(g['cs%d' % i]), flip=bool(par[0]),
              sat_first=bool(par[1]), delta=float(par[2]), factor=float(par[3]), training=bool(par[4]), cropping=bool(par[5]),
              crop=crop, moff=moff)


def test_preprocess_oracle_hits_the_modes_fixture():
  """--do_random_cropping, RANDOM_CROP / NONE and the colour spaces: the restatement against what the reference's own
  preprocess_image computed when the fixture was made (tools/make_golden.py --preprocess-modes)."""
  g = np.load(MODES)
  seen = set()
  for i in range(int(g['n'])):
    c = _mode_case(g, i)
    got = N.preprocess_image(c['img'], int(g['hw']), c['mode'], c['training'], flip=c['flip'], saturation_first=c['sat_first'],
                             delta=c['delta'], factor=c['factor'], crop=c['crop'], color_space=c['cs'], mode_offset=c['moff'])
    assert np.abs(got - c['want']).max() < 1e-12, i
    assert (c['crop'] is not None) == (c['cropping'] and c['training'])      # an evaluation call ignores the flag
    seen.add((c['mode'], c['cs'], c['crop'] is not None))
  assert {m for m, _, _ in seen} == {'PAD', 'CROP', 'RESHAPE', 'RANDOM_CROP', 'NONE'} and {c for _, c, _ in seen} == {'rgb', 'yiq', 'bgr', 'gray'}


@pytest.mark.skipif(not os.path.isdir('/root/reference'), reason='the reference tree is only mounted in the build container')
def test_preprocess_modes_fixture_is_what_the_reference_computes():
  """Re-executes preprocessing/danbooru_preprocessing.preprocess_image on the stand-in for every fixture case: same
  draws, same output."""
  from oracle import ref_runner
  sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
  import make_golden
  g = np.load(MODES)
  assert int(g['n']) == len(make_golden.PREPROCESS_MODE_CASES)
  for i, (h, w, mode, training, seed, cropping, cs) in enumerate(make_golden.PREPROCESS_MODE_CASES):
    c = _mode_case(g, i)
    assert c['img'].shape == (h, w, 3) and (c['mode'], c['cs'], c['training'], c['cropping']) == (mode, cs, training, cropping)
    res, dr = ref_runner.run_preprocess(c['img'], int(g['hw']), mode, training, seed, do_random_cropping=cropping, color_space=cs)
    assert np.abs(res - c['want']).max() == 0.0 and dr['crop'] == c['crop']
    assert (None if dr['mode_crop'] is None else tuple(dr['mode_crop'][:2])) == c['moff']


def test_crop_draws_and_resize_modes():
  """draw_crops: sizes int32(mid * U[0.8, 1)) per axis, offsets inside the intermediate image (preprocessing_util.py:312-331);
  source_rect for RANDOM_CROP / NONE (:84-95,137-139); the packed tables carry the crop table only for a training call."""
  gen = np.random.default_rng(1)
  crop = D.draw_crops(5000, 320, 0.8, gen)
  assert crop.dtype == np.int32 and crop[:, 2:].min() == 256 and crop[:, 2:].max() == 319
  assert (crop[:, :2] >= 0).all() and (crop[:, :2] + crop[:, 2:] <= 320).all() and (crop[:, 0] + crop[:, 2]).max() == 320
  assert abs(np.corrcoef(crop[:, 2], crop[:, 3])[0, 1]) < 0.05                   # height and width have their own draws
  assert D.source_rect(20, 45, 'RANDOM_CROP', 40, gen) == (0, 0, 20, 45)         # smaller than the target: resized whole
  for _ in range(50):
    oy, ox, sh, sw = D.source_rect(64, 70, 'RANDOM_CROP', 40, gen)
    assert (sh, sw) == (40, 40) and 0 <= oy <= 24 and 0 <= ox <= 30
  assert D.source_rect(64, 70, 'RANDOM_CROP', 40, None, offset=(24, 30)) == (24, 30, 40, 40)
  assert D.source_rect(32, 32, 'NONE', 32) == (0, 0, 32, 32)
  with pytest.raises(ValueError):
    D.source_rect(32, 31, 'NONE', 32)
  with pytest.raises(ValueError):
    D.source_rect(32, 32, 'STRETCH', 32)
  imgs = [np.zeros((50, 60, 3), np.uint8), np.zeros((41, 40, 3), np.uint8)]
  pre = D.Preprocessor(32, device='cpu', resize_mode='RESHAPE', do_random_cropping=True)
  assert pre.crops and pre.mid == 40 and len(pre.pack(imgs)) == 5
  tables = pre.pack(imgs, crop=np.array([[0, 0, 40, 40], [3, 4, 33, 36]]))
  assert tables[4].dtype == torch.int32 and tables[4].tolist() == [[0, 0, 40, 40], [3, 4, 33, 36]]
  with pytest.raises(AssertionError):
    pre.pack(imgs, crop=np.array([[0, 0, 41, 40], [3, 4, 33, 36]]))
  ev = D.Preprocessor(32, device='cpu', resize_mode='RESHAPE', do_random_cropping=True, is_training=False)
  assert not ev.crops and len(ev.pack(imgs)) == 4                               # danbooru_preprocessing.py:187-190
  assert [D.Preprocessor(int(hw), device='cpu', do_random_cropping=True).mid for hw in (4, 8, 16, 32, 64, 128, 256, 512)] == \
      [5, 10, 20, 40, 80, 160, 320, 640]
  with pytest.raises(AssertionError):
    D.Preprocessor(32, device='cpu', color_space='hsv')


def test_embedding_dataset_records_and_fields(tmp_path):
  """datasets/celeba_facenet.py:86-118: an image + 'image/embedding' FixedLenFeature; decode() hands the vector over as the
  item 'embedding'; a record of the wrong length is an error; the batch stacker gives [batch, size] fp32; unknown dataset
  names are refused as dataset_factory.py:77-78 does."""
  from PIL import Image
  rng = np.random.RandomState(5)
  recs, embs, imgs = [], [], []
  for i in range(6):
    a = rng.randint(0, 256, (12 + i, 10, 3), dtype=np.uint8)
    buf = io.BytesIO()
    Image.fromarray(a).save(buf, format='PNG')
    e = rng.randn(8).astype(np.float32)
    recs.append(D.embedding_example(buf.getvalue(), e, 'png', 'f%d' % i, **{'image/attribs': [1, 0, 1]}))
    embs.append(e)
    imgs.append(a)
  recs.append(D.embedding_example(buf.getvalue(), rng.randn(7), 'png', 'short'))
  D.write_tfrecords(str(tmp_path / 'train-00000-of-00001'), recs)
  ds = D.EmbeddingImageDataset(str(tmp_path), 'train', embedding_size=8)
  assert 'celeba_facenet' in D.DATASETS and D.EMBEDDING_SIZE == 512
  payloads = list(ds.records())
  decoded = [ds.decode(p) for p in payloads[:6]]
  for i, (im, name, fields) in enumerate(decoded):
    np.testing.assert_array_equal(im, imgs[i])
    assert name == 'f%d' % i and fields['embedding'].dtype == np.float32
    np.testing.assert_array_equal(fields['embedding'], embs[i])      # float32 survives the tf.Example float list bit-exactly
  with pytest.raises(ValueError):
    ds.decode(payloads[6])
  stacked = D._stack_fields(decoded[:4])
  assert set(stacked) == {'embedding'} and tuple(stacked['embedding'].shape) == (4, 8) and stacked['embedding'].dtype == torch.float32
  assert D._stack_fields([(imgs[0], 'x')]) is None                      # an image-only dataset has no further fields
  two = D.TwoDomainBatches(str(tmp_path), str(tmp_path), device='cpu', dataset_names=('celeba_facenet', 'svhn'))
  with pytest.raises(ValueError):
    two._dataset(1, str(tmp_path))
  assert isinstance(two._dataset(0, str(tmp_path)), D.EmbeddingImageDataset)


def _tables_to_image(img, rect, mid, crop, hw):
  """What tg_preprocess_images_crop computes from its tables (include/twingan_hip.h), restated with the oracle's resize:
  the source rectangle (zero-padded outside the image) -> [mid, mid] (or straight to [hw, hw] without a crop table) ->
  the crop rectangle -> [hw, hw].  Lets the host-side table logic be checked on CPU against the reference's results."""
  h, w, y0, x0, sh, sw = (int(v) for v in rect)
  x = (img.astype(np.float32) * np.float32(1.0 / 255.0)).astype(np.float64)
  src = np.zeros((sh, sw, 3))
  ys, xs = max(y0, 0), max(x0, 0)
  ye, xe = min(y0 + sh, h), min(x0 + sw, w)
  src[ys - y0:ye - y0, xs - x0:xe - x0] = x[ys:ye, xs:xe]
  if crop is None:
    return N.resize_bilinear_tf1(src, hw, hw)
  cy, cx, ch, cw = (int(v) for v in crop)
  return N.resize_bilinear_tf1(N.resize_bilinear_tf1(src, mid, mid)[cy:cy + ch, cx:cx + cw], hw, hw)


@pytest.mark.skipif(not os.path.isdir('/root/reference'), reason='the reference tree is only mounted in the build container')
def test_host_tables_reproduce_the_live_reference_for_every_resize_mode():
  """The tables Preprocessor.pack builds (source rectangle, intermediate size, crop rectangle) fed through a restatement of
  the kernel's table semantics equal the reference's own resize_image / random_crop_image for every mode, evaluation calls
  (no colour ops), with the reference's random draws: PAD, CROP, RESHAPE, RANDOM_CROP, RANDOM_CROP_AND_RESHAPE
  (--random_crop_and_reshape_initial_crop_hw; preprocessing_util.py:24-27,128-131), with and without --do_random_cropping."""
  from oracle import ref_runner
  rng = np.random.RandomState(31)
  cases = [(37, 53, 'PAD', False, None), (64, 40, 'CROP', False, None), (20, 33, 'RESHAPE', True, None),
           (60, 70, 'RANDOM_CROP', False, None), (64, 70, 'RANDOM_CROP', True, None), (20, 45, 'RANDOM_CROP', True, None),
           (60, 70, 'RANDOM_CROP_AND_RESHAPE', False, 48), (30, 45, 'RANDOM_CROP_AND_RESHAPE', False, 48),
           (50, 41, 'RANDOM_CROP_AND_RESHAPE', False, 41)]
  for i, (h, w, mode, cropping, c) in enumerate(cases):
    img = rng.randint(0, 256, (h, w, 3), dtype=np.uint8)
    # a TRAINING call with flip / colour switched off is not available in the reference: compare the geometry on an
    # evaluation call when no cropping is asked for, else undo nothing -- feed gray (no colour ops) and un-flip
    training = cropping
    res, dr = ref_runner.run_preprocess(img, 32, mode, training, seed=200 + i, do_random_cropping=cropping,
                                        color_space='gray', initial_crop_hw=c)
    if training and dr['flip_uniform'] < 0.5:
      res = res[:, ::-1]
    pre = D.Preprocessor(32, device='cpu', resize_mode=mode, is_training=training, do_random_cropping=cropping,
                         color_space='gray', initial_crop_hw=c)
    moff = None if dr['mode_crop'] is None else tuple(dr['mode_crop'][:2])
    tables = pre.pack([img], crop=None if dr['crop'] is None else np.array([dr['crop']]), mode_offsets=[moff])
    rect = tables[2][0].tolist()
    crop = tables[4][0].tolist() if len(tables) == 5 else None
    got = _tables_to_image(img, rect, pre.mid, crop, 32)
    assert np.abs(got - res).max() < 1e-12, (i, mode, np.abs(got - res).max())
  with pytest.raises(AssertionError):
    D.Preprocessor(32, device='cpu', resize_mode='RANDOM_CROP_AND_RESHAPE')                        # needs the flag
  with pytest.raises(AssertionError):
    D.Preprocessor(32, device='cpu', resize_mode='RANDOM_CROP_AND_RESHAPE', initial_crop_hw=24)    # up-sampling window
  with pytest.raises(AssertionError):
    D.Preprocessor(32, device='cpu', resize_mode='RANDOM_CROP_AND_RESHAPE', initial_crop_hw=48, do_random_cropping=True)


def test_labelled_datasets_are_read_for_their_images(tmp_path):
  """'anime_faces' -- the target domain of the reference's own TwinGAN recipe (docs/training.md:16-17) -- and 'celeba' carry
  tags / attributes / landmarks next to the image (datasets/anime_faces.py:77-88, celeba.py:82-92); the trainer reads the
  image only.  get_dataset opens them by their '<split>-*' pattern and skips the other features."""
  from PIL import Image
  rng = np.random.RandomState(6)
  recs, imgs = [], []
  for i in range(3):
    a = rng.randint(0, 256, (9, 7 + i, 3), dtype=np.uint8)
    buf = io.BytesIO()
    Image.fromarray(a).save(buf, format='PNG')
    recs.append(D.encode_example({'image/encoded': buf.getvalue(), 'image/format': 'png', 'image/filename': 'face%d' % i,
                                  'image/class/label': [3, 17], 'image/class/text': 'blue hair, smile'}))
    imgs.append(a)
  D.write_tfrecords(str(tmp_path / 'train-00000-of-00001'), recs)
  D.write_tfrecords(str(tmp_path / 'trainextra'), recs[:1])      # matches image_only's '%s*' pattern, not '%s-*'
  for name in ('anime_faces', 'celeba'):
    ds = D.get_dataset(name, 'train', str(tmp_path))
    assert [os.path.basename(f) for f in ds.files] == ['train-00000-of-00001']
    got = [ds.decode(p) for p in ds.records()]
    assert [g[1] for g in got] == ['face0', 'face1', 'face2'] and all(len(g) == 2 for g in got)
    for g, a in zip(got, imgs):
      np.testing.assert_array_equal(g[0], a)
  assert len(D.get_dataset('image_only', 'train', str(tmp_path)).files) == 2
  with pytest.raises(ValueError):
    D.get_dataset('svhn', 'train', str(tmp_path))
