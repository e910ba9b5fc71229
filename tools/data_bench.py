"""Throughput of the input pipeline on the GPU box: synthetic 256x256 JPEGs in TFRecord files -> Loader -> device
batches (decode on host threads, resize / flip / colour distortion in tg_preprocess_images).  Prints images/s for the
whole pipeline and for the GPU preprocessing alone."""
import io
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from twingan_amd import data as D  # noqa: E402


def main(hw=256, n_images=2048, batch=64, workers=(8,)):
  from PIL import Image
  d = tempfile.mkdtemp()
  rng = np.random.RandomState(0)
  recs = []
  for i in range(256):
    yy, xx = np.mgrid[0:hw + 44, 0:hw + 12]
    a = np.stack([(yy * (1 + i % 3) + 7 * i) % 256, (xx * 2 + i) % 256, ((yy + xx) // 2) % 256], axis=-1).astype(np.uint8)
    a = np.clip(a.astype(int) + rng.randint(-12, 12, a.shape), 0, 255).astype(np.uint8)
    buf = io.BytesIO()
    Image.fromarray(a).save(buf, format='JPEG', quality=90)
    recs.append(D.image_example(buf.getvalue(), 'jpeg', '%d' % i))
  per = n_images // 8
  for f in range(8):
    D.write_tfrecords(os.path.join(d, 'train-%05d-of-00008' % f), [recs[(f * per + k) % len(recs)] for k in range(per)])
  print('dataset: %d records, %.1f KB per JPEG' % (n_images, np.mean([len(r) for r in recs]) / 1e3), flush=True)
  ds = D.ImageOnlyDataset(d, 'train')
  # GPU preprocessing alone
  imgs = [D.decode_image(D.decode_example(r)['image/encoded'][0]) for r in recs[:batch]]
  pre = D.Preprocessor(hw, device='cuda:0', precision='bf16')
  packed = pre.pack(imgs)
  for _ in range(3):
    pre.run(*packed)
  torch.cuda.synchronize()
  t0 = time.time()
  for _ in range(20):
    pre.run(*packed)
  torch.cuda.synchronize()
  print('upload + preprocess kernel: %.0f images/s' % (20 * batch / (time.time() - t0)), flush=True)
  t0 = time.time()
  for _ in range(5):
    pre.pack(imgs)
  print('host packing: %.0f images/s per thread' % (5 * batch / (time.time() - t0)), flush=True)
  t0 = time.time()
  for r in recs[:64]:
    D.decode_image(D.decode_example(r)['image/encoded'][0])
  print('JPEG decode: %.0f images/s per thread' % (64 / (time.time() - t0)), flush=True)
  for w in workers:
    ld = D.Loader(ds, batch, D.Preprocessor(hw, device='cuda:0', precision='bf16'), num_readers=4, num_workers=w, seed=1)
    try:
      for _ in range(3):
        ld.next()
      torch.cuda.synchronize()
      t0 = time.time()
      nb = 24
      for _ in range(nb):
        ld.next()
      torch.cuda.synchronize()
      print('loader, %d decode threads: %.0f images/s' % (w, nb * batch / (time.time() - t0)), flush=True)
    finally:
      ld.close()
  for pcs in (16, 48):
    ld = D.Loader(ds, batch, D.Preprocessor(hw, device='cuda:0', precision='bf16'), processes=pcs, seed=1, pool=batch * pcs)
    try:
      for _ in range(2 * pcs // 8 + 3):
        ld.next()
      torch.cuda.synchronize()
      t0 = time.time()
      nb = 100
      for _ in range(nb):
        ld.next()
      torch.cuda.synchronize()
      print('loader, %d decode processes: %.0f images/s' % (pcs, nb * batch / (time.time() - t0)), flush=True)
    finally:
      ld.close()


if __name__ == '__main__':
  main()
