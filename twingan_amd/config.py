"""The reference flags that shape the hot path, as one dataclass (the reference scatters them over
tf.flags in nets/pggan.py:24-59, image_generation.py:50-121, twingan.py:39-88,
model/model_inheritor.py:41-304)."""
from dataclasses import dataclass


@dataclass
class Config:
  hw: int = 256                       # --train_image_size (pggan_runner.py:136-150)
  max_ch: int = 256                   # --pggan_max_num_channels           nets/pggan.py:51-53
  max_ch_dis: object = None           # --pggan_max_num_channels_dis (nets/pggan.py:54-56): discriminators only; None = max_ch
  # nets/pggan.py:24, nets/pggan_utils.py:35-41: instance_norm (north star) | batch_norm | batch_renorm | none |
  # batch_renorm_native | layer_norm_native (tf.contrib's own layers; TwinGAN trainer only)
  generator_norm_type: str = 'instance_norm'
  do_pixel_norm: bool = True          # nets/pggan.py:34-38
  use_unet: bool = True               # twingan.py:53-56
  unet_max_concat_hw: object = None   # --pggan_unet_max_concat_hw (nets/pggan.py:57-59): no UNet skip above this hw
  equalized_learning_rate: bool = False   # nets/pggan.py:39-41; nets/pggan_utils.py:82-84,236-254
  use_res_block: bool = False         # nets/pggan.py:43-46; nets/pggan_utils.py:257-264,334-342
  use_larger_filter_at_rgb_layer: bool = False   # nets/pggan.py:47-50: 7x7 (min(7, hw/2)) to-RGB kernels
  spectral_norm: bool = False         # nets/pggan.py:28-30; libs/sn.py:38-101 (discriminator convs)
  spectral_norm_in_non_discriminator: bool = False   # nets/pggan.py:31-33
  do_self_attention: bool = False     # image_generation.py:62-64; libs/self_attention.py:24-70
  self_attention_hw: int = 64         # image_generation.py:65-67
  use_style_embedding: bool = False   # twingan.py:47-49: generator norm parameters conditioned on a style embedding
  style_embed_size: int = 16          # twingan.py:50-51
  do_encoder_distillation: bool = False   # twingan.py:58-65: the content encoder distils dataset-provided embeddings
  distillation_weight: float = 1.0
  distillation_start_hw: int = 16
  distill_embed_dim: int = 0          # width of the dataset's 'a_embedding' / 'b_embedding' fields (twingan.py:164-177)
  # gdrop (libs/gdrop.py:20-36).  As in the reference: --use_gdrop creates the `gdrop_strength` variable and its controller
  # (image_generation.py:563-585, 1034-1040) and hands the variable to the discriminators, but the layer itself runs only
  # under nets/pggan.py's `do_dgrop` argument, which no trainer of the reference sets (default False, :340,352)
  use_gdrop: bool = False
  gdrop_coef: float = 0.2
  gdrop_lim: float = 0.5
  gdrop_exp: float = 2.0
  do_dgrop: bool = False              # nets/pggan.py:340 (the reference's spelling)
  gdrop_strength: float = 0.0         # nets/pggan.py:341: the strength when no `gdrop_strength` variable exists
  is_training: bool = True            # False: the inference branch (twingan.py:300-363) -- BatchNorm reads the moving statistics
  is_growing: bool = False            # image_generation.py:69-72
  alpha_grow: float = 0.0             # twingan.py:833-835
  loss_architecture: str = 'wgan_gp'  # image_generation.py:81-83
  gradient_penalty_lambda: float = 10.0   # image_generation.py:92-95
  gan_weight: float = 1.0             # image_generation.py:84-86
  wgan_drift_loss_weight: float = 0.0     # image_generation.py:96-98
  l_cyc_weight: float = 1.0           # twingan.py:73-76
  l_content_weight: float = 0.1       # twingan.py:80-82
  do_l_cyc_gan: bool = True           # twingan.py:77-79
  n_critic: int = 2                   # image_generation.py:87-90
  learning_rate: float = 1e-4         # docs/training.md:24-25
  use_ttur: bool = False              # image_generation.py:554-561; accepted, changes no update (the reference applies the
  discriminator_learning_rate: float = 4e-4   # discriminator gradients with the generator's optimizer, :640-646)
  adam_beta1: float = 0.5
  adam_beta2: float = 0.99
  opt_epsilon: float = 1e-8
  precision: str = 'bf16'             # 'bf16' | 'fp16': 16-bit activations + MFMA convs, fp32 master weights ('fp16' = the
                                      # reference's --dataset_dtype float16: set loss_scale, 128 there); 'fp32': exact path
  domain_streams: bool = True         # run the two (independent) discriminators on two HIP streams
  overlap_cut_hw: int = 32            # data-parallel runs: the backward is cut where the feature maps grow past this size and
                                      # the all-reduce of the (large) lower-resolution gradients overlaps the rest of it
  loss_scale: float = 1.0             # --mix_precision_loss_scale (model_inheritor.py:568-570); bf16 needs none
