"""Single-image translation on the training kernels -- the MI355X counterpart of inference/image_translation_infer.py
(:46-99): load a stage's variables, map uint8 images to [0, 1] floats of the model's resolution
(tf.image.convert_image_dtype + tf.image.resize_images, :57-59), run the inference branch of the TwinGAN graph
(twingan.translate = `sources_ph -> custom_generated_t_style_source`, twingan.py:300-363,777-805) and return images
scaled by 255 (:87).  Image file IO (util_io.imread / the web demo) is out of scope: arrays in, arrays out."""
import numpy as np
import torch

from .config import Config
from .params import ParamStore, declare_twingan
from .twingan import encode_style, translate

OUTPUT_TENSORS = tuple('custom_generated_%s_style_%s' % (d, k) for d in 'st' for k in ('rand', 'source', 'target', 'ph'))


def resize_bilinear_tf1(img, hw):
  """tf.image.resize_images(method=BILINEAR, align_corners=False) of TF 1.8 for [B, H, W, C] float tensors: source
  coordinate = destination index * (in / out) (no half-pixel offset), the two neighbours clamped at the border."""
  b, h, w, c = img.shape
  if (h, w) == (hw, hw):
    return img

  def axis(n_in, n_out):
    pos = torch.arange(n_out, device=img.device, dtype=torch.float32) * (float(n_in) / float(n_out))
    lo = pos.floor().clamp_(0, n_in - 1)
    hi = (lo + 1).clamp_(max=n_in - 1)
    return lo.long(), hi.long(), (pos - lo)
  y0, y1, fy = axis(h, hw)
  x0, x1, fx = axis(w, hw)
  fy = fy.view(1, hw, 1, 1)
  fx = fx.view(1, 1, hw, 1)
  top = img[:, y0][:, :, x0] * (1 - fx) + img[:, y0][:, :, x1] * fx
  bot = img[:, y1][:, :, x0] * (1 - fx) + img[:, y1][:, :, x1] * fx
  return top * (1 - fy) + bot * fy


class ImageInferer:
  """ImageInferer of inference/image_translation_infer.py:46-99 without the TF session: variables come from a state
  dict keyed by the reference's variable names (ParamStore.state_dict(include_state=True), or a converted checkpoint)."""

  def __init__(self, cfg, state_dict, device='cuda', output_tensor_name='custom_generated_t_style_source'):
    assert output_tensor_name in OUTPUT_TENSORS, output_tensor_name
    self.cfg = cfg if isinstance(cfg, Config) else Config(**cfg)
    self.device = torch.device(device)
    self.store = declare_twingan(ParamStore(self.device), self.cfg).build(0)
    self.store.load_state_dict(state_dict)
    to, self.style_from = output_tensor_name[len('custom_generated_'):].split('_style_')
    self.to = to
    self.dtype = {'bf16': torch.bfloat16, 'fp16': torch.float16, 'fp32': torch.float32}[self.cfg.precision]

  @classmethod
  def from_checkpoint(cls, cfg, model_path, device='cuda', output_tensor_name='custom_generated_t_style_source'):
    """``model_path``: a TF-format checkpoint prefix or the train_dir that holds one (image_translation_infer.py:60-74
    restores tf.train.latest_checkpoint(model_path)): the model's variables are read by name (checkpoint.py)."""
    import os
    from . import checkpoint as ckpt
    cfg = cfg if isinstance(cfg, Config) else Config(**cfg)
    prefix = ckpt.latest_checkpoint(model_path) if os.path.isdir(model_path) else model_path
    if prefix is None:
      raise FileNotFoundError('no checkpoint in %s' % model_path)
    probe = declare_twingan(ParamStore(torch.device('cpu')), cfg).build(0)
    names = set(probe.specs) | set(probe.state_specs)
    probe.close()
    arrays = ckpt.read_checkpoint(prefix, names=names)
    missing = sorted(names - set(arrays))
    if missing:
      raise KeyError('checkpoint %s lacks %d variable(s) of this configuration, e.g. %s' % (prefix, len(missing), missing[0]))
    return cls(cfg, {k: torch.from_numpy(v.astype(np.float32)) for k, v in arrays.items()}, device, output_tensor_name)

  def preprocess(self, images):
    """uint8 [H,W,3] / [B,H,W,3] (or floats already in [0,1]) -> device tensor [B, hw, hw, 3] of the model's dtype."""
    x = torch.as_tensor(np.asarray(images))
    if x.dim() == 3:
      x = x.unsqueeze(0)
    x = x.to(self.device)
    x = x.float() / 255.0 if x.dtype == torch.uint8 else x.float()      # tf.image.convert_image_dtype
    return resize_bilinear_tf1(x, self.cfg.hw).to(self.dtype).contiguous()

  def infer(self, images, style_embed=None):
    """-> float32 numpy [B, hw, hw, 3], range 0..255 (image_translation_infer.py:85-88)."""
    x = self.preprocess(images)
    style = None
    with torch.cuda.device(self.device):
      if self.cfg.use_style_embedding:
        frm = 's' if self.to == 't' else 't'
        if self.style_from == 'ph':
          assert style_embed is not None, 'custom_generated_*_style_ph needs a style embedding'
          style = torch.as_tensor(style_embed, dtype=torch.float32, device=self.device)
        elif self.style_from == 'rand':
          style = torch.randn(x.shape[0], self.cfg.style_embed_size, device=self.device)
        else:      # the style of the input image itself ('source' for s->t, 'target' for t->s)
          style = encode_style(self.store.P, x, self.cfg, frm)
      out = translate(self.store.P, x, self.cfg, self.to, style)
    return (out.float() * 255.0).cpu().numpy()
