"""The preprocessing kernel (tg_preprocess_images, csrc/preprocess.hip) against the reference's own preprocess_image
(tests/golden/preprocess_hw32.npz: executed on the TF stand-in when the fixture was made) and the float64 oracle, and the
loader end to end on a synthetic TFRecord dataset."""
import io
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import np_ops as N          # noqa: E402  (checker only)

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'preprocess_hw32.npz')


def _cases():
  g = np.load(GOLD)
  out, i = [], 0
  while 'img%d' % i in g:
    out.append((g['img%d' % i], str(g['mode%d' % i]), g['par%d' % i], g['out%d' % i]))
    i += 1
  return out, int(g['hw'])


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_preprocess_kernel_hits_the_reference_fixture(precision):
  """One launch per resize mode over images of different sizes; fp32 output within 2e-6 of the reference's float64
  result (fp32 arithmetic of ~10 operations on values in [0, 1]), bf16 within one rounding (2^-8)."""
  from twingan_amd import data as D
  cases, hw = _cases()
  for mode in ('PAD', 'CROP', 'RESHAPE'):
    sel = [c for c in cases if c[1] == mode]
    pre = D.Preprocessor(hw, device='cuda:0', precision=precision, resize_mode=mode)
    aug = np.stack([c[2][:4] for c in sel]).astype(np.float32)
    out = pre([c[0] for c in sel], aug=aug).float().cpu().numpy()
    for k, c in enumerate(sel):
      err = np.abs(out[k] - c[3]).max()
      assert err < (2e-6 if precision == 'fp32' else 2.0 ** -8), (mode, k, err)


def test_preprocess_kernel_matches_oracle_on_large_images():
  """256x256 targets from larger, non-square sources (the bench resolution), random draws from the host helper."""
  from twingan_amd import data as D
  rng = np.random.RandomState(11)
  imgs = [rng.randint(0, 256, (h, w, 3), dtype=np.uint8) for h, w in ((300, 420), (512, 384), (256, 256), (199, 611))]
  pre = D.Preprocessor(256, device='cuda:0', precision='fp32', resize_mode='PAD', seed=3)
  aug = D.draw_augmentation(len(imgs), np.random.default_rng(5))
  out = pre(imgs, aug=aug).cpu().numpy()
  for k, im in enumerate(imgs):
    want = N.preprocess_image(im, 256, 'PAD', True, flip=bool(aug[k, 0]), saturation_first=bool(aug[k, 1]),
                              delta=float(aug[k, 2]), factor=float(aug[k, 3]))
    assert np.abs(out[k] - want).max() < 2e-6, k
  ev = D.Preprocessor(256, device='cuda:0', precision='fp32', resize_mode='PAD', is_training=False)(imgs).cpu().numpy()
  for k, im in enumerate(imgs):
    assert np.abs(ev[k] - N.preprocess_image(im, 256, 'PAD', False)).max() < 2e-6


def test_loader_end_to_end(tmp_path):
  """TFRecord files of JPEG tf.Examples -> shuffled, decoded, preprocessed device batches; in evaluation mode without
  shuffling the batch equals the oracle's preprocessing of the decoded images, in order."""
  from PIL import Image
  from twingan_amd import data as D
  rng = np.random.RandomState(2)
  recs, decoded = [], []
  for i in range(24):
    h, w = 40 + 3 * (i % 5), 64 - 2 * (i % 7)
    yy, xx = np.mgrid[0:h, 0:w]
    a = np.stack([(3 * yy + i * 9) % 256, (2 * xx + 5 * i) % 256, (yy + xx) % 256], axis=-1).astype(np.uint8)
    buf = io.BytesIO()
    Image.fromarray(a).save(buf, format='JPEG', quality=90)
    recs.append(D.image_example(buf.getvalue(), 'jpeg', 'im%02d' % i))
    decoded.append(D.decode_image(buf.getvalue()))
  D.write_tfrecords(str(tmp_path / 'train-00000-of-00002'), recs[:12])
  D.write_tfrecords(str(tmp_path / 'train-00001-of-00002'), recs[12:])
  ds = D.ImageOnlyDataset(str(tmp_path), 'train')
  ev = D.Loader(ds, 8, D.Preprocessor(32, device='cuda:0', precision='fp32', is_training=False), num_readers=1,
                num_workers=1, shuffle=False)
  try:
    for b in range(3):
      batch = ev.next().cpu().numpy()
      assert batch.shape == (8, 32, 32, 3)
      for k in range(8):
        want = N.preprocess_image(decoded[b * 8 + k], 32, 'PAD', False)
        assert np.abs(batch[k] - want).max() < 2e-6, (b, k)
  finally:
    ev.close()
  tr = D.Loader(ds, 8, D.Preprocessor(32, device='cuda:0', precision='bf16'), num_readers=2, num_workers=3, seed=1)
  try:
    seen = []
    for _ in range(6):
      batch = tr.next()
      assert batch.dtype == torch.bfloat16 and batch.shape == (8, 32, 32, 3)
      assert float(batch.min()) >= 0 and float(batch.max()) <= 1
      seen.append(batch.float().mean().item())
    assert len(set(round(v, 4) for v in seen)) > 1      # shuffled, augmented batches differ
  finally:
    tr.close()


def test_loader_with_decode_processes(tmp_path):
  """Loader(processes=P): worker processes decode and pack, the consumer uploads and preprocesses; evaluation mode
  without shuffling, one process: batches equal the oracle's, in order."""
  from PIL import Image
  from twingan_amd import data as D
  recs, decoded = [], []
  for i in range(16):
    h, w = 36 + 2 * (i % 3), 48 + i
    yy, xx = np.mgrid[0:h, 0:w]
    a = np.stack([(5 * yy + i) % 256, (3 * xx + 2 * i) % 256, (yy * 2 + xx) % 256], axis=-1).astype(np.uint8)
    buf = io.BytesIO()
    Image.fromarray(a).save(buf, format='PNG')
    recs.append(D.image_example(buf.getvalue(), 'png', '%d' % i))
    decoded.append(a)
  D.write_tfrecords(str(tmp_path / 'train-00000-of-00001'), recs)
  ds = D.ImageOnlyDataset(str(tmp_path), 'train')
  ld = D.Loader(ds, 4, D.Preprocessor(24, device='cuda:0', precision='fp32', is_training=False), shuffle=False, processes=1)
  try:
    for b in range(4):
      batch = ld.next().cpu().numpy()
      for k in range(4):
        assert np.abs(batch[k] - N.preprocess_image(decoded[b * 4 + k], 24, 'PAD', False)).max() < 2e-6, (b, k)
  finally:
    ld.close()
  tr = D.Loader(ds, 4, D.Preprocessor(24, device='cuda:0', precision='bf16'), processes=2, pool=8, seed=3)
  try:
    for _ in range(5):
      batch = tr.next()
      assert batch.shape == (4, 24, 24, 3) and float(batch.min()) >= 0 and float(batch.max()) <= 1
  finally:
    tr.close()
  # the worker processes rebuild the preprocessor: the RANDOM_CROP_AND_RESHAPE window must travel with it
  rc = D.Loader(ds, 4, D.Preprocessor(16, device='cuda:0', precision='bf16', resize_mode='RANDOM_CROP_AND_RESHAPE',
                                      initial_crop_hw=24), processes=1, pool=4, seed=5)
  try:
    batch = rc.next()
    assert batch.shape == (4, 16, 16, 3) and float(batch.min()) >= 0 and float(batch.max()) <= 1
  finally:
    rc.close()
  # a dead decode worker raises in next() instead of blocking the consumer forever
  dead = D.Loader(ds, 4, D.Preprocessor(24, device='cuda:0', precision='bf16'), processes=1, pool=4, seed=6)
  try:
    dead.next()
    for pr in dead.procs:
      pr.terminate()
      pr.join(timeout=5.0)
    with pytest.raises(RuntimeError, match='decode worker'):
      for _ in range(64):      # whatever was already queued is still served
        dead.next()
  finally:
    dead.close()


def test_progressive_training_from_tfrecord_datasets(tmp_path):
  """End to end on files: two image-only TFRecord datasets (the unpaired source / target domains) -> loaders ->
  GPU preprocessing -> runner.run_progressive 4 -> 8 with checkpoints per stage; losses stay finite, the loaders are
  rebuilt per stage resolution, the final stage directory holds a TF-format checkpoint that ImageInferer loads."""
  from PIL import Image
  from twingan_amd import Config, checkpoint as C, data as D
  from twingan_amd.inference import ImageInferer
  from twingan_amd.runner import run_progressive
  rng = np.random.RandomState(4)
  for dom in ('a', 'b'):
    recs = []
    for i in range(12):
      base = rng.randint(0, 256, (1, 1, 3))
      yy, xx = np.mgrid[0:40, 0:36]
      a = np.clip(base + np.stack([yy * 3, xx * 4, yy + xx], axis=-1) * (1 if dom == 'a' else -1), 0, 255).astype(np.uint8)
      buf = io.BytesIO()
      Image.fromarray(a).save(buf, format='PNG')
      recs.append(D.image_example(buf.getvalue(), 'png', '%s%d' % (dom, i)))
    os.makedirs(tmp_path / dom)
    D.write_tfrecords(str(tmp_path / dom / 'train-00000-of-00001'), recs)
  batches = D.TwoDomainBatches(str(tmp_path / 'a'), str(tmp_path / 'b'), device='cuda:0', precision='bf16', num_workers=2, seed=1)
  seen = []

  def batch_fn(hw, bsz):
    s, t = batches(hw, bsz)
    seen.append((hw, tuple(s.shape), tuple(t.shape)))
    return s, t
  try:
    state, hist = run_progressive(Config(hw=4, max_ch=16, precision='bf16'), batch_fn, 4, 8, {4: 4, 8: 4},
                                  num_images_per_resolution=8, device='cuda:0', seed=2, max_steps_per_stage=2,
                                  train_dir=str(tmp_path / 'run'))
  finally:
    batches.close()
  assert [h['stage'] for h in hist] == ['4', '4to8', '8']
  assert {s[0] for s in seen} == {4, 8} and all(s[1] == (4, s[0], s[0], 3) for s in seen)
  assert all(torch.isfinite(v).all() for v in state.values())
  last = C.latest_checkpoint(str(tmp_path / 'run' / '8'))
  assert last is not None and last.endswith('model.ckpt-2')
  inf = ImageInferer.from_checkpoint(Config(hw=8, max_ch=16, precision='bf16', is_training=False), str(tmp_path / 'run' / '8'),
                                     device='cuda:0')
  out = inf.infer(rng.randint(0, 256, (2, 20, 24, 3), dtype=np.uint8))
  assert out.shape == (2, 8, 8, 3) and np.isfinite(out).all()


MODES = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'preprocess_modes_hw32.npz')


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_preprocess_modes_hit_the_reference_fixture(precision):
  """--do_random_cropping (the reference's training recipe, docs/training.md:22-23), the RANDOM_CROP / NONE resize modes
  and --color_space yiq / bgr / gray: tg_preprocess_images_crop against what the reference's own preprocess_image computed
  for the same draws (tests/golden/preprocess_modes_hw32.npz, tools/make_golden.py --preprocess-modes).  The two chained
  bilinear resizes run inside one launch (16 source taps per output pixel): fp32 within 3e-6, bf16 within one rounding of
  the value's magnitude (yiq reaches beyond [0, 1])."""
  from twingan_amd import data as D
  g = np.load(MODES)
  hw = int(g['hw'])
  for i in range(int(g['n'])):
    img, want, par = g['img%d' % i], g['out%d' % i], g['par%d' % i]
    mode, cs, training, cropping = str(g['mode%d' % i]), str(g['cs%d' % i]), bool(par[4]), bool(par[5])
    crop = g['crop%d' % i] if (training and cropping) else None
    moff = tuple(int(v) for v in g['moff%d' % i]) if g['moff%d' % i][0] >= 0 else None
    pre = D.Preprocessor(hw, device='cuda:0', precision=precision, resize_mode=mode, is_training=training,
                         do_random_cropping=cropping, color_space=cs)
    assert pre.crops == (crop is not None) and (not pre.crops or pre.mid == 40)
    out = pre([img], aug=par[None, :4].astype(np.float32) if training else None, crop=None if crop is None else crop[None],
              mode_offsets=[moff]).float().cpu().numpy()[0]
    err = np.abs(out - want).max()
    assert err < (3e-6 if precision == 'fp32' else 2.0 ** -8 * max(1.0, np.abs(want).max())), (i, mode, cs, err)


def test_random_cropping_at_the_bench_resolution_matches_oracle():
  """256 x 256 targets with --resize_mode=RESHAPE --do_random_cropping=True from larger sources: the host's own draws
  (draw_crops on the 320 x 320 intermediate), a batch in one launch, against the float64 oracle; and the loader hands the
  crop table through its worker threads."""
  from twingan_amd import data as D
  rng = np.random.RandomState(12)
  imgs = [rng.randint(0, 256, (h, w, 3), dtype=np.uint8) for h, w in ((300, 420), (512, 384), (256, 256), (199, 611), (700, 500))]
  pre = D.Preprocessor(256, device='cuda:0', precision='fp32', resize_mode='RESHAPE', seed=4, do_random_cropping=True)
  assert pre.mid == 320
  gen = np.random.default_rng(6)
  aug, crop = D.draw_augmentation(len(imgs), gen), D.draw_crops(len(imgs), 320, 0.8, gen)
  assert (crop[:, 2:] >= 256).all() and (crop[:, 2:] < 320).all() and (crop[:, :2] + crop[:, 2:] <= 320).all()
  out = pre(imgs, aug=aug, crop=crop).cpu().numpy()
  for k, im in enumerate(imgs):
    want = N.preprocess_image(im, 256, 'RESHAPE', True, flip=bool(aug[k, 0]), saturation_first=bool(aug[k, 1]),
                              delta=float(aug[k, 2]), factor=float(aug[k, 3]), crop=tuple(crop[k]))
    assert np.abs(out[k] - want).max() < 3e-6, k
  own = pre(imgs)      # every draw from the preprocessor's stream
  assert own.shape == (5, 256, 256, 3) and float(own.min()) >= 0.0 and float(own.max()) <= 1.0


def test_distillation_trains_from_an_embedding_dataset(tmp_path):
  """--do_encoder_distillation end to end on files (twingan.py:103,162-177,507-521): the source domain is a
  celeba_facenet-style dataset (image + 'image/embedding', datasets/celeba_facenet.py:86-99), the target an image-only
  one; the loader hands the embeddings of exactly the images of the batch to the trainer (evaluation order checked
  against the file), run_progressive passes them on as distill_embed_s, and the distillation terms of the source side
  (and only those) appear in the loss."""
  from PIL import Image
  from twingan_amd import Config, data as D
  from twingan_amd.runner import run_progressive
  from twingan_amd.twingan import Trainer
  rng = np.random.RandomState(7)
  embs = rng.randn(12, 6).astype(np.float32)
  for dom in ('a', 'b'):
    recs = []
    for i in range(12):
      yy, xx = np.mgrid[0:24, 0:20]
      a = np.clip(np.stack([yy * 7 + i, xx * 9, yy + xx + 11 * i], axis=-1) % 256, 0, 255).astype(np.uint8)
      buf = io.BytesIO()
      Image.fromarray(a).save(buf, format='PNG')
      recs.append(D.embedding_example(buf.getvalue(), embs[i], 'png', 'a%d' % i) if dom == 'a'
                  else D.image_example(buf.getvalue(), 'png', 'b%d' % i))
    os.makedirs(tmp_path / dom)
    D.write_tfrecords(str(tmp_path / dom / 'train-00000-of-00001'), recs)
  # the loader keeps image and embedding together: evaluation order, no shuffling
  ds = D.EmbeddingImageDataset(str(tmp_path / 'a'), 'train', embedding_size=6)
  ev = D.Loader(ds, 4, D.Preprocessor(16, device='cuda:0', precision='fp32', is_training=False), num_readers=1, num_workers=1,
                shuffle=False)
  try:
    for b in range(3):
      images, fields = ev.next()
      assert images.shape == (4, 16, 16, 3) and fields['embedding'].is_cuda
      np.testing.assert_array_equal(fields['embedding'].cpu().numpy(), embs[b * 4:b * 4 + 4])
  finally:
    ev.close()
  batches = D.TwoDomainBatches(str(tmp_path / 'a'), str(tmp_path / 'b'), device='cuda:0', precision='bf16', num_workers=2, seed=1,
                               dataset_names=('celeba_facenet', 'image_only'), embedding_size=6, resize_mode='RESHAPE',
                               do_random_cropping=True)
  cfg = Config(hw=16, max_ch=16, precision='bf16', do_encoder_distillation=True, distill_embed_dim=6, distillation_weight=0.5)
  try:
    s, t, extras = batches(16, 4)
    assert set(extras) == {'distill_embed_s'} and tuple(extras['distill_embed_s'].shape) == (4, 6)
    rows = {tuple(np.round(r, 5)) for r in embs}
    assert all(tuple(np.round(r, 5)) in rows for r in extras['distill_embed_s'].cpu().numpy())
    tr = Trainer(cfg, device='cuda:0', seed=3)
    loss, terms = tr.run(s, t, **extras)
    assert {'l_source_distillation', 'l_t_prime_distillation'} <= set(terms) and 'l_target_distillation' not in terms
    assert 0.0 <= float(terms['l_source_distillation']) <= 2.0 * 0.5      # weight * (1 - cos) per image, averaged
    tr.close()
    state, hist = run_progressive(cfg, batches, 16, 16, {16: 4}, num_images_per_resolution=8, device='cuda:0', seed=2,
                                  max_steps_per_stage=2)
  finally:
    batches.close()
  assert [h['stage'] for h in hist] == ['16'] and all(torch.isfinite(v).all() for v in state.values())


def test_random_crop_and_reshape_matches_oracle():
  """--resize_mode=RANDOM_CROP_AND_RESHAPE with --random_crop_and_reshape_initial_crop_hw (preprocessing_util.py:24-27,
  128-131): a random [c, c] window (a smaller image is first resized up to [c, c]), then the resize to hw -- the kernel's
  two-stage path with the intermediate size c.  Evaluation and training calls against the float64 oracle (itself held to
  the reference live: tests/test_data_cpu.py::test_host_tables_reproduce_the_live_reference_for_every_resize_mode)."""
  from twingan_amd import data as D
  rng = np.random.RandomState(14)
  imgs = [rng.randint(0, 256, (h, w, 3), dtype=np.uint8) for h, w in ((60, 70), (30, 45), (48, 48), (100, 49))]
  offs = [(9, 16), None, (0, 0), (37, 1)]      # None: the image is smaller than the window and resized whole
  for training in (False, True):
    pre = D.Preprocessor(32, device='cuda:0', precision='fp32', resize_mode='RANDOM_CROP_AND_RESHAPE', initial_crop_hw=48,
                         is_training=training, seed=2)
    aug = D.draw_augmentation(len(imgs), np.random.default_rng(8)) if training else None
    out = pre(imgs, aug=aug, mode_offsets=offs).cpu().numpy()
    for k, im in enumerate(imgs):
      kw = dict(flip=bool(aug[k, 0]), saturation_first=bool(aug[k, 1]), delta=float(aug[k, 2]), factor=float(aug[k, 3])) \
          if training else {}
      want = N.preprocess_image(im, 32, 'RANDOM_CROP_AND_RESHAPE', training, mode_offset=offs[k] or (0, 0), initial_crop_hw=48,
                                **kw)
      assert np.abs(out[k] - want).max() < 3e-6, (training, k, np.abs(out[k] - want).max())
  own = D.Preprocessor(32, device='cuda:0', precision='bf16', resize_mode='RANDOM_CROP_AND_RESHAPE', initial_crop_hw=40)(imgs)
  assert own.shape == (4, 32, 32, 3) and float(own.min()) >= 0.0 and float(own.max()) <= 1.0
