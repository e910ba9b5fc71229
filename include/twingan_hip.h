/*
 * twingan_hip.h -- C ABI of libtwingan_hip.so: the MI355X (gfx950) kernels behind the TwinGAN
 * G+D training hot path.
 *
 * The reference (jerryli27/TwinGAN) has no FFI: its "operator API" for this path is the Python op
 * facade libs/ops.py:28-40 + nets/pggan_utils.py maybe_* helpers, which dispatch to stock TF-1.8
 * kernels.  Each entry point below names the reference call site (file:line under /root/reference)
 * whose TF kernel(s) it replaces.  twingan_amd/_lib.py is the ctypes binding; INTEGRATION.md shows
 * the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - all activation tensors are NHWC, C-contiguous, 16-byte aligned base pointers;
 *   - `dtype` selects the activation storage type: TG_F32, TG_BF16 or TG_F16; accumulation is always fp32;
 *   - conv weights are TF HWIO [kh][kw][cin][cout] fp32 ("master") unless a parameter says "packed";
 *   - the caller owns all memory; the library never allocates device memory and never synchronises;
 *     kernels are enqueued on `stream` (a hipStream_t passed as void*; NULL = the null stream);
 *   - return value: 0 = success, negative TG_E* = failure; tg_last_error() returns a thread-local
 *     message.  Nothing is enqueued when an error is returned.
 */
#ifndef TWINGAN_HIP_H_
#define TWINGAN_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TG_F32 0
#define TG_BF16 1
#define TG_F16 2            /* IEEE half storage (the reference's --dataset_dtype float16 with loss scale 128,
                               deployment/model_deploy.py:146-183,308-313); fp32 accumulation and master weights */

#define TG_OK 0
#define TG_EINVAL (-1)   /* bad shape / dtype / flag combination */
#define TG_EALIGN (-2)   /* pointer or channel count not aligned as required */
#define TG_ELAUNCH (-3)  /* hipLaunchKernel reported an error */
#define TG_ENOSUP (-4)   /* combination not implemented by the selected algorithm */
#define TG_ECOMM (-5)    /* RCCL reported an error (tg_comm_*, tg_allreduce); tg_last_error() carries its text */

/* Conv algorithm selector */
#define TG_ALGO_DIRECT 0 /* one thread per output, any shape, f32 or bf16 activations */
#define TG_ALGO_MFMA 1   /* LDS-tiled implicit GEMM on v_mfma_f32_32x32x16_bf16, bf16 activations only */
#define TG_ALGO_MFMA_V1 2 /* same weight packs, forces the first-generation MFMA kernels (A/B benchmarking) */

/* Epilogue / prologue flags */
#define TG_EPI_BIAS 1
#define TG_EPI_LRELU 2

int tg_version(void);
const char* tg_last_error(void);
/* name of the kernel variant the last conv dispatch on this thread selected (e.g. "conv_tile_kernel<3,32,64,1>");
 * lets host-side timing be attributed to the kernel symbols rocprofv3 reports */
const char* tg_last_kernel(void);
/* Deterministic mode (process-wide; default = TG_DETERMINISTIC in the environment, 0 if unset).  The fp32 storage type
 * always sums in a fixed order; with the mode on, the 16-bit storage types do too: every sum that ends in one number per
 * channel / sample / tensor is taken by one workgroup or from per-workgroup partials in a fixed order instead of fp32
 * atomics in arrival order, so two runs of a step are bit-identical (TF's graph has no such switch: the reference's
 * cuDNN / Eigen reductions are whatever the device scheduled, model/model_inheritor.py:537-571 trains without seeds for
 * them).  Returns the previous setting.  Do not flip it between capturing and replaying a hipGraph. */
int tg_set_deterministic(int on);
int tg_get_deterministic(void);
/* Deferred filter-gradient reductions (process-wide).  Between tg_wgrad_defer(1) and tg_wgrad_defer_flush(stream) the
 * split-K slab reduction of every filter gradient that ACCUMULATES into a caller buffer (tg_conv2d_bwd_weight*,
 * tg_conv2d_upcat_bwd_weight with accumulate != 0) is queued instead of launched, and the flush issues all queued
 * reductions as ONE launch on `stream` (up to 120 per flush; a full queue falls back to the immediate launch; the
 * deterministic mode never defers).  The caller must keep the workspaces of those calls alive until the flush and flush on
 * a stream that is ordered after all of them -- and before anything reads the gradients.  tg_wgrad_defer returns the
 * previous setting; tg_wgrad_defer(0) drops what was not flushed.  tg_wgrad_defer_flush returns the number of reductions
 * it issued.  The reference has no counterpart (TF schedules its gradient ops itself). */
int tg_wgrad_defer(int on);
int tg_wgrad_defer_flush(void* stream);
/* Fixed-order ("ordered") forms of the sums whose plain entry points end in fp32 atomics: every workgroup writes its
 * partial result to a row of `workspace` (fp32, workspace_floats long: 512 rows are enough for any size; fewer rows mean
 * fewer workgroups) and one pass adds the rows in workgroup order -- the same bits on every run, at full-grid speed, in
 * either mode and for every storage type.  twingan_amd/ops.py routes the bias gradients, the fromRGB / toRGB filter
 * gradients and the loss sums through these when the deterministic mode is on.
 *   tg_channel_sum_ordered:  out[c] (+)= sum over pixels of g[.., c]                (BiasAddGrad)
 *   tg_sum_ordered:          out[0] (+)= scale * sum x   (y NULL)  or  scale * sum |x - y|
 *   tg_pointwise_conv_bwd_weight_ordered: as tg_pointwise_conv_bwd_weight */
int tg_channel_sum_ordered(const void* g, float* out, int64_t npix, int c, int accumulate, float* workspace,
                           size_t workspace_floats, int dtype, void* stream);
int tg_sum_ordered(const void* x, const void* y_or_null, float* out, int64_t numel, float scale, int accumulate,
                   float* workspace, size_t workspace_floats, int dtype, void* stream);
int tg_pointwise_conv_bwd_weight_ordered(const void* x, const void* gy, float* gw, int64_t npix, int cin, int cout,
                                         int accumulate, float* workspace, size_t workspace_floats, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Convolution: stride 1, kh,kw <= 4, arbitrary zero padding (pad_t/pad_l on the low side; the
 * high side follows from hout/wout).  Replaces tf.contrib.layers.conv2d -> Conv2D /
 * Conv2DBackpropInput / Conv2DBackpropFilter at nets/pggan_utils.py:316-320 (every E/G/D conv in
 * nets/pggan.py) and the 4x4 VALID convs at nets/pggan.py:153,330,495.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  int32_t n, hin, win, cin;   /* input  [n,hin,win,cin]  (cin = physical channels)           */
  int32_t hout, wout, cout;   /* output [n,hout,wout,cout]                                   */
  int32_t kh, kw, pad_t, pad_l;
  int32_t dtype;              /* TG_F32 | TG_BF16 (activations x, y, gy, gx)                 */
  int32_t algo;               /* TG_ALGO_*                                                   */
  int32_t epilogue;           /* TG_EPI_* bits (forward only)                                */
  float lrelu_alpha;          /* util_misc.py:68 (0.2)                                       */
  int32_t groups;             /* 0 / 1: one weight set.  G > 1 (<= TG_MAX_GROUPS, n % G == 0): the batch is G equal image
                                 ranges and range g is convolved with weight set g -- `w` (master [G][kh][kw][cin][cout] or
                                 G packs back to back: tg_conv2d_pack_elems counts all of them), `bias` [G][cout], `gw`, `gbias`
                                 hold G sets; every other operand is the plain [n, ...] tensor.  The reference builds
                                 discriminator_s and discriminator_t as two towers of identical layers (twingan.py:105-110,
                                 image_generation.py:348-439 per domain): G = 2 runs a layer of both as one launch.  Entry
                                 points whose kernel cannot select the set per image launch once per group. */
} TgConvDesc;
#define TG_MAX_GROUPS 4

/* y = epilogue(conv(x, w)).  DIRECT: w = fp32 HWIO master (rounded to bf16 on read when dtype is
 * bf16).  MFMA: w = bf16 pack from tg_conv2d_pack_weights(mode 0).  bias: fp32 [cout] or NULL. */
int tg_conv2d_fwd(const TgConvDesc* d, const void* x, const void* w, const float* bias, void* y, void* stream);

/* gx = d conv / d x applied to gy (Conv2DBackpropInput).  `d` is the FORWARD descriptor.
 * DIRECT: w = fp32 HWIO master.  MFMA: w = bf16 pack from tg_conv2d_pack_weights(mode 1). */
int tg_conv2d_bwd_data(const TgConvDesc* d, const void* gy, const void* w, void* gx, void* stream);
/* The same with the LeakyReLU backward of the layer that PRODUCED this conv's input folded in:
 * gx = bwd_data(gy) * (x_act > 0 ? 1 : d->lrelu_alpha), x_act = the forward input [n,hin,win,cin] (an activation
 * output, so x_act > 0 iff its pre-activation was).  One epilogue read of x_act instead of a separate
 * read-read-write pass (tf LeakyReluGrad after Conv2DBackpropInput). */
int tg_conv2d_bwd_data_masked(const TgConvDesc* d, const void* gy, const void* w, const void* x_act, void* gx,
                              void* stream);
/* Backward-data of the LAST conv of a discriminator block (nets/pggan.py:304-306: conv -> LeakyReLU -> tf.nn.avg_pool) taken
 * straight from the gradient of the POOLED output: the conv's incoming gradient
 *   gy[n,y,x,c] = rnd(0.25 * gy_pooled[n,y/2,x/2,c] * (bit c of y_signs[n,y,x] ? 1 : d->lrelu_alpha))
 * (AvgPoolGrad + LeakyReluGrad, what tg_lrelu_pool_bwd_signs writes) is formed while the kernel stages its input tiles and
 * is never in memory.  gy_pooled [n,hout/2,wout/2,cout]; y_signs [n,hout,wout,cout/8] bytes from tg_conv2d_fwd_pool_signs;
 * x_act (may be NULL): as in tg_conv2d_bwd_data_masked; w: pack of mode 1.  gx is bit-identical to the two-launch path.
 * gy_out (may be NULL): [n,hout,wout,cout] -- the kernel also WRITES gy (each element once, from the tile that owns the pixel),
 * for the layer's filter / bias gradient in a discriminator step: the tg_lrelu_pool_bwd_signs launch and this kernel's read of
 * its output are still gone.  tg_conv2d_bwd_data_unpool_supported(d) != 0: the layer is one the kernels take (3x3 SAME on the
 * tile kernels' maps, cout % 32 == 0, 16-bit storage); callers keep the two-launch path otherwise. */
int tg_conv2d_bwd_data_unpool_supported(const TgConvDesc* d);
int tg_conv2d_bwd_data_unpool(const TgConvDesc* d, const void* gy_pooled, const void* y_signs, const void* w, const void* x_act,
                              void* gx, void* gy_out, void* stream);
/* The same for a pass that kept the layer's activation output y_act [n,hout,wout,cout] instead of its sign bytes (the
 * gradient-penalty pass, image_generation.py:414-439, whose second differentiation reads y_act): bit j = (y_act > 0),
 * i.e. gy = what tg_lrelu_pool_bwd writes from (gy_pooled, y_act). */
int tg_conv2d_bwd_data_unpool_act(const TgConvDesc* d, const void* gy_pooled, const void* y_act, const void* w, const void* x_act,
                                  void* gx, void* gy_out, void* stream);
/* The adjoint of that node, as the gradient penalty's second backward pass needs it (image_generation.py:414-439: the
 * backward of tf.gradients(pred, interp)): y = conv(x, w) * (mask_src > 0 ? 1 : d->lrelu_alpha) with mask_src [n,hout,
 * wout,cout] -- the forward conv of the incoming cotangent with the LeakyReLU mask of the NEXT node of that pass in its
 * epilogue (one launch instead of conv + LeakyReluGrad).  d->epilogue must be 0; w as in tg_conv2d_fwd. */
int tg_conv2d_fwd_masked(const TgConvDesc* d, const void* x, const void* w, const void* mask_src, void* y, void* stream);

/* gw (fp32 HWIO) = d conv / d w (Conv2DBackpropFilter), `d` is the FORWARD descriptor.
 * workspace: tg_conv2d_bwd_weight_workspace(d) bytes (split-K partial slabs; may be 0/NULL).
 * accumulate != 0 adds into gw instead of overwriting it. */
size_t tg_conv2d_bwd_weight_workspace(const TgConvDesc* d);
int tg_conv2d_bwd_weight(const TgConvDesc* d, const void* x, const void* gy, float* gw, int accumulate,
                         void* workspace, size_t workspace_bytes, void* stream);

/* The same filter gradient over TWO batches of one layer in one launch: gw (+)= wgrad(xa [d->n images], gya) +
 * wgrad(xb [nb images], gyb) -- e.g. the batched real/fake/interpolate pass and the gradient-penalty double-backward
 * term of a discriminator conv, or the two encoder passes of a generator step.  `d` describes batch a.
 * tg_conv2d_bwd_weight2_workspace returns 0 when the layer is not eligible (then issue two ordinary calls). */
size_t tg_conv2d_bwd_weight2_workspace(const TgConvDesc* d, int nb);
int tg_conv2d_bwd_weight2(const TgConvDesc* d, int nb, const void* xa, const void* gya, const void* xb, const void* gyb,
                          float* gw, int accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* Filter gradient AND bias gradient of a conv + bias layer from one read of gy: gbias[cout] += sum over pixels of gy
 * (BiasAddGrad; always ADDS -- the caller zeroes).  The tile kernels get it from one extra MFMA per K step with an
 * all-ones operand; other shapes fall back to tg_channel_sum. */
int tg_conv2d_bwd_weight_bias(const TgConvDesc* d, const void* x, const void* gy, float* gw, float* gbias, int accumulate,
                              void* workspace, size_t workspace_bytes, void* stream);
/* bias_segs: bit 0 = batch a contributes to gbias, bit 1 = batch b (the gradient-penalty double-backward term of a
 * layer carries no bias gradient). */
int tg_conv2d_bwd_weight2_bias(const TgConvDesc* d, int nb, const void* xa, const void* gya, const void* xb,
                               const void* gyb, float* gw, float* gbias, int bias_segs, int accumulate, void* workspace,
                               size_t workspace_bytes, void* stream);

/* generator_three_layer_block's first conv (nets/pggan.py:69-78) with its input concat(nearest_up2(x0), x1)
 * (resize_twice_as_big + maybe_concat_unet_layer, nets/pggan_utils.py:281-298,349-350) read straight from the two
 * sources instead of from a materialised copy: y[n,h,w,cout] = conv3x3_same(concat(up2(x0 [n,h/2,w/2,c0]),
 * x1 [n1,h,w,c1]), w) and its filter gradient gw[3][3][c0+c1][cout].  16-bit activations (dtype = TG_BF16 / TG_F16, packs
 * in the same format); w_pack = mode-0 pack of the
 * ordinary [3,3,c0+c1,cout] kernel; (gsz, perm) as in tg_upsample2x_concat_fwd.  The input gradient is
 * tg_conv2d_upcat_bwd_data (or the ordinary tg_conv2d_bwd_data followed by tg_upsample2x_concat_bwd).  tg_conv2d_upcat_supported: h % 8 == 0, w % 16 == 0,
 * c0 % 32 == 0, c1 % 32 == 0, cout % 8 == 0. */
int tg_conv2d_upcat_supported(int h, int w, int c0, int c1, int cout);
int tg_conv2d_upcat_fwd(const void* x0, const void* x1, const void* w_pack, void* y, int n, int h, int w, int c0, int c1,
                        int cout, int gsz, unsigned perm, int dtype, void* stream);
/* The same conv's input gradient written straight into the gradients of the two sources (Conv2DBackpropInput +
 * ConcatV2 / ResizeNearestNeighbor gradients of nets/pggan_utils.py:281-298,349-350 in one launch): g0 [n,h/2,w/2,c0] =
 * 2x2 sums of channels [0,c0) of conv3x3^T(gy [n,h,w,cout], w), g1 [n1,h,w,c1] = channels [c0,c0+c1), summed over the groups
 * that read one skip image ((gsz, perm) as above; n1 = (max perm + 1) * gsz, or n without groups) -- from the fp32
 * accumulators, one rounding; the concat-layout gradient tensor is never written.  w_pack = mode-1 pack of the kernel for
 * the descriptor (n,h,w,c0+c1) -> cout.  Either output may be NULL (not wanted).  Shapes: tg_conv2d_upcat_supported. */
int tg_conv2d_upcat_bwd_data(const void* gy, const void* w_pack, void* g0, void* g1, int n, int h, int w, int c0, int c1,
                             int cout, int gsz, unsigned perm, int dtype, void* stream);
/* Conv followed by a normaliser (every encoder / generator conv: layers.conv2d(normalizer_fn=instance_norm),
 * nets/pggan_utils.py:86-98 -> tf.nn.moments over the conv output, libs/instance_norm.py:131): the forward conv also
 * writes, per output channel, the sum and the sum of squares of the (bf16-rounded) outputs each workgroup produced --
 * partials fp32 [n][chunks][2][cout], unshifted, summed in a fixed order -- so that the moments need no second read of
 * y (tg_norm_act_fwd_conv_stats consumes them).  tg_conv2d_fwd_stats_chunks: chunks per image for this descriptor, or 0
 * when the kernel it dispatches has no statistics epilogue (the caller then uses tg_instance_norm_partials).  Plain
 * epilogue only (d->epilogue == 0: normalised convs have no bias), 3x3, MFMA path. */
int tg_conv2d_fwd_stats_chunks(const TgConvDesc* d);
/* Last conv of a discriminator block together with the tf.nn.avg_pool that follows it (nets/pggan.py:304-306): y as
 * tg_conv2d_fwd (bias / LeakyReLU epilogue as d->epilogue says) and y_pooled [n, h/2, w/2, cout] = the 2x2 average of
 * the bf16-rounded y, written from the same output tile.  tg_conv2d_fwd_pool_supported: 3x3 SAME, h % 8 == 0,
 * w % 16 == 0, MFMA path. */
int tg_conv2d_fwd_pool_supported(const TgConvDesc* d);
int tg_conv2d_fwd_pool(const TgConvDesc* d, const void* x, const void* w_pack, const float* bias, void* y, void* y_pooled,
                       void* stream);
/* The same conv + pool when the pool is the ONLY consumer of the full-resolution output (the discriminator blocks,
 * nets/pggan.py:304-306: `net` is overwritten by the pooled tensor): all the backward pass needs of y is the LeakyReLU
 * derivative, i.e. the sign of each element -- y itself is not written, y_signs [n][h][w][cout/8] bytes is (bit j of
 * byte q = (y[.., 8q+j] > 0), taken from the storage-rounded value exactly as tg_conv2d_fwd_pool would have stored it):
 * 1/16 of the tensor's bytes, once, instead of a write and a read of all of it.  cout % 8 == 0; y_signs 4-byte aligned.
 * Consumed by tg_lrelu_pool_bwd_signs.  Same shapes as tg_conv2d_fwd_pool_supported. */
int tg_conv2d_fwd_pool_signs(const TgConvDesc* d, const void* x, const void* w_pack, const float* bias, void* y_signs,
                             void* y_pooled, void* stream);
int tg_conv2d_fwd_stats(const TgConvDesc* d, const void* x, const void* w_pack, void* y, float* partials, int chunks,
                        void* stream);
int tg_conv2d_upcat_fwd_stats_chunks(int n, int h, int w, int c0, int c1, int cout);
int tg_conv2d_upcat_fwd_stats(const void* x0, const void* x1, const void* w_pack, void* y, float* partials, int chunks,
                              int n, int h, int w, int c0, int c1, int cout, int gsz, unsigned perm, int dtype, void* stream);
size_t tg_conv2d_upcat_bwd_weight_workspace(int n, int h, int w, int c0, int c1, int cout);
int tg_conv2d_upcat_bwd_weight(const void* x0, const void* x1, const void* gy, float* gw, int accumulate, void* workspace,
                               size_t workspace_bytes, int n, int h, int w, int c0, int c1, int cout, int gsz,
                               unsigned perm, int dtype,
                               void* stream);

/* bf16 K-contiguous weight packs for the MFMA kernels, from the fp32 HWIO master (`d` = forward
 * descriptor; a kxk VALID conv on a kxk input is packed as the equivalent dense 1x1 over k*k*cin).
 * mode 0 (forward):  out[co][tap][ci]        = w[tap][ci][co]
 * mode 1 (bwd data): out[ci][tap'][co]       = w[ntaps-1-tap'][ci][co]   (180-degree rotated taps)
 * rows (first index) are padded to a multiple of 64 and the innermost index to a multiple of 16,
 * zero filled.  tg_conv2d_pack_elems returns the element count of the pack. */
size_t tg_conv2d_pack_elems(const TgConvDesc* d, int mode);
int tg_conv2d_pack_weights(const TgConvDesc* d, const float* w_hwio, int mode, void* out_bf16, void* stream);
/* A pack is OPAQUE: made for (d, mode), read by the conv entry point called with the same d.  For the layers whose
 * dispatch ends in the whole-image kernel (3x3 over 8x8 maps) the elements are stored in MFMA-fragment order instead
 * ([row / 32][k / 16][tap][row % 32][k % 16]: one contiguous KB per wave load).  tg_conv2d_pack_layout tells which
 * (0 = the [row][tap][k] order above, 1 = fragment order): a host that caches packs by (weight, mode) must not share one
 * between descriptors whose layouts differ (twingan_amd/ops.py PackCache keys on it). */
int tg_conv2d_pack_layout(const TgConvDesc* d, int mode);

/* All packs of an optimiser group in ONE launch (the step re-packs ~60 weights after every Adam apply; one
 * launch per pack is launch-latency bound).  The caller builds a job table in HOST memory with
 * tg_pack_table_fill (job j of njobs; *total_blocks accumulates the grid size, start it at 0), copies its
 * tg_pack_table_bytes(njobs) bytes to the device once, and calls tg_conv2d_pack_weights_multi every step. */
size_t tg_pack_table_bytes(int njobs);
int tg_pack_table_fill(const TgConvDesc* d, const float* w_hwio, int mode, void* out_bf16, int job, void* table_host,
                       int32_t* total_blocks);
int tg_conv2d_pack_weights_multi(const void* table_device, int njobs, int total_blocks, void* stream);

/* ---------------------------------------------------------------------------------------------
 * 1x1 convs with a 3-channel side (fromRGB 3->C, toRGB C->3): pure-bandwidth VALU kernels.
 * nets/pggan.py:233-240,395-399 (from_rgb), :176-178,198-200 (to_rgb).  w = fp32 [cin][cout].
 * y = epilogue(x @ w).  bwd_data of one is the forward of the other with w transposed (wt != 0
 * reads w as [cout][cin]).  bwd_weight: gw[cin][cout] = sum_pixels x^T gy.
 * ------------------------------------------------------------------------------------------- */
int tg_pointwise_conv_fwd(const void* x, const float* w, const float* bias, void* y, int64_t npix, int cin, int cout,
                          int wt, int epilogue, float lrelu_alpha, int dtype, void* stream);
/* y = rnd(x @ w) * (mask > 0 ? 1 : alpha), mask of y's shape, cin <= 4: the transposed toRGB-side conv of the gradient
 * penalty's second backward with the LeakyReluGrad of fromRGB's output in its epilogue (image_generation.py:414-439). */
int tg_pointwise_conv_fwd_masked(const void* x, const float* w, const void* mask, void* y, int64_t npix, int cin, int cout,
                                 int transpose_w, float alpha, int dtype, void* stream);
int tg_pointwise_conv_bwd_weight(const void* x, const void* gy, float* gw, int64_t npix, int cin, int cout,
                                 int accumulate, int dtype, void* stream);
/* The fromRGB layer's filter AND bias gradient from one read of gy (Conv2DBackpropFilter + BiasAddGrad of
 * nets/pggan.py:233-240 in the discriminator arg-scope, nets/pggan_utils.py:116-127): gw as above, gbias[cout] (+)= sum
 * over pixels of gy.  cin <= 4.  One launch at the shapes the RGB kernel takes (16-bit, cin 3, npix % 4 == 0), else the
 * filter gradient followed by tg_channel_sum. */
int tg_pointwise_conv_bwd_weight_bias(const void* x, const void* gy, float* gw, float* gbias, int64_t npix, int cin,
                                      int cout, int accumulate, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Instance norm (+ LeakyReLU + pixel norm), libs/instance_norm.py:131-135, util_misc.py:68-86,
 * nets/pggan_utils.py:330-331.  Layer order: conv -> norm -> act -> pixel-norm (pggan.py:78-81).
 * ------------------------------------------------------------------------------------------- */
/* mean[n*c], rstd[n*c] (fp32) over (h,w) of y[n,h,w,c]; biased variance, rstd = rsqrt(var+eps). */
int tg_instance_norm_stats(const void* y, float* mean, float* rstd, int n, int h, int w, int c, float eps, int dtype,
                           void* stream);
/* z = pixnorm(lrelu((y-mean)*rstd*gamma + beta)); flags: bit0 lrelu, bit1 pixel-norm.
 * gamma/beta fp32 [c].  Per-domain parameters (conditional_layer_var_scope_postfix '_s' / '_t',
 * nets/pggan_utils.py:102-113): when gamma2/beta2 are non-NULL, images [0, split) use gamma/beta and images
 * [split, n) use gamma2/beta2, so the source- and target-domain passes of one network run as ONE batch.
 * pn_scale (fp32 [n*h*w], may be NULL unless pixel-norm) receives 1/sqrt(mean_c(a^2)+pn_eps) for the backward. */
/* per_image_params != 0: gamma / beta are [n][c] -- one row per image (batch renorm: every pass batched along n has
 * its own r*gamma, d*gamma+beta); gamma2 / beta2 / split are then ignored. */
int tg_norm_act_fwd(const void* y, const float* mean, const float* rstd, const float* gamma, const float* beta,
                    const float* gamma2, const float* beta2, int split, int per_image_params, void* z, float* pn_scale,
                    int n, int h, int w, int c, int flags, float lrelu_alpha, float pn_eps, int dtype, void* stream);
/* The same forward as two launches instead of four (no zero fill, no atomics, no finalise kernel):
 * tg_instance_norm_partials writes per-block shifted sums to partials (fp32 [n * tg_norm_chunks(n,h,w) * 2 * c]);
 * tg_norm_act_fwd_partials finalises mean / rstd from them in its prologue, WRITES mean[n*c] / rstd[n*c] (for the
 * backward and the BatchNorm moving averages) and applies the affine + LeakyReLU + pixel norm.  z_pooled (may be NULL):
 * also receives the 2x2 average pool of z, [n, h/2, w/2, c] (tf.nn.avg_pool after an encoder block, pggan.py:466-468). */
int tg_norm_chunks(int n, int h, int w);
int tg_instance_norm_partials(const void* y, float* partials, int n, int h, int w, int c, int dtype, void* stream);
int tg_norm_act_fwd_partials(const void* y, const float* partials, float* mean, float* rstd, const float* gamma,
                             const float* beta, const float* gamma2, const float* beta2, int split, void* z,
                             void* z_pooled, float* pn_scale, int n, int h, int w, int c, int flags, float lrelu_alpha,
                             float in_eps, float pn_eps, int dtype, void* stream);
/* tg_norm_act_fwd_partials fed by the partial sums of the producing conv (tg_conv2d_fwd_stats /
 * tg_conv2d_upcat_fwd_stats: [n][part_chunks][2][c], unshifted). */
int tg_norm_act_fwd_conv_stats(const void* y, const float* partials, int part_chunks, float* mean, float* rstd,
                               const float* gamma, const float* beta, const float* gamma2, const float* beta2, int split,
                               void* z, void* z_pooled, float* pn_scale, int n, int h, int w, int c, int flags,
                               float lrelu_alpha, float in_eps, float pn_eps, int dtype, void* stream);
/* Backward of tg_norm_act_fwd.  Inputs: gz [n,h,w,c] and/or gz_pooled [n,h/2,w/2,c] (either may be NULL; the
 * layer-output gradient is gz + 0.25 * upsample(gz_pooled): the tf.nn.avg_pool that follows an encoder block,
 * nets/pggan.py:436,468, is folded in), y (raw conv output), pn_scale, mean, rstd, gamma, beta (+ 2nd domain).
 * Outputs: gy (same dtype), ggamma[c], gbeta[c] (+ ggamma2, gbeta2 for images >= split) (fp32, may be NULL;
 * accumulate != 0 adds; with per_image_params ggamma / gbeta are [n][c] and written).  sums: fp32 scratch [n * tg_norm_chunks(n,h,w) * 2 * c] (per-block partial sums). */
int tg_norm_act_bwd(const void* gz, const void* gz_pooled, const void* y, const float* pn_scale, const float* mean,
                    const float* rstd, const float* gamma, const float* beta, const float* gamma2, const float* beta2, int split, void* gy,
                    int per_image_params, float* ggamma, float* gbeta, float* ggamma2, float* gbeta2, float* sums, int n,
                    int h, int w, int c, int flags, float lrelu_alpha, int accumulate, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Discriminator pointwise: bias + LeakyReLU (nets/pggan_utils.py:116; slim BiasAdd) and pieces
 * of its (double) backward.
 * ------------------------------------------------------------------------------------------- */
/* z = lrelu(y + bias) (bias may be NULL; alpha = 1 disables the activation) */
int tg_bias_lrelu_fwd(const void* y, const float* bias, void* z, int64_t npix, int c, float alpha, int dtype,
                      void* stream);
/* gy = gz * (z > 0 ? 1 : alpha)   (also its own double backward wrt gz) */
int tg_lrelu_bwd(const void* gz, const void* z, void* gy, int64_t numel, float alpha, int dtype, void* stream);
/* gy = gz * (z > 0 ? 1 : alpha) and gbias[c] (+)= sum over pixels of gy  -- LeakyReLU backward fused with
 * BiasAddGrad (one pass over gz, z instead of two kernels) */
int tg_lrelu_bwd_bias(const void* gz, const void* z, void* gy, float* gbias, int64_t npix, int c, float alpha,
                      int accumulate, int dtype, void* stream);
/* Same with the 2x2 average pool that follows a discriminator block (nets/pggan.py:274,306) folded in: the
 * incoming gradient is gz [n,h,w,c] (may be NULL) + 0.25 * upsample(gz_pooled [n,h/2,w/2,c]) (may be NULL);
 * gbias may be NULL (no bias gradient wanted). */
int tg_lrelu_pool_bwd(const void* gz, const void* gz_pooled, const void* z, void* gy, float* gbias, int n, int h, int w,
                      int c, float alpha, int accumulate, int dtype, void* stream);
/* tg_lrelu_pool_bwd for a layer run by tg_conv2d_fwd_pool_signs: gy = 0.25 * upsample2(gz_pooled) * (sign bit ? 1 : alpha),
 * gbias += sum over pixels of gy (NULL: no bias gradient).  16-bit storage, c % 8 == 0. */
int tg_lrelu_pool_bwd_signs(const void* gz_pooled, const void* z_signs, void* gy, float* gbias, int n, int h, int w, int c,
                            float alpha, int accumulate, int dtype, void* stream);
/* out[c] (fp32) = sum over pixels of g[pix][c]  (BiasAddGrad) */
int tg_channel_sum(const void* g, float* out, int64_t npix, int c, int accumulate, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Resampling / concat / fade-in.
 * ------------------------------------------------------------------------------------------- */
/* out[n,2h,2w,c0+c1] = concat(nearest_up2(x0[n,h,w,c0]), x1[n',2h,2w,c1]); c1 == 0 -> plain upsample.
 * nets/pggan_utils.py:349-350 + :281-298 (generator features first).
 * gsz == 0: n' = n, image i reads skip image i.  gsz > 0: the n images are n/gsz (<= 4) groups of gsz; output
 * group k reads skip group (perm >> 8k) & 0xff -- the four generator passes batched along N read the UNet skips
 * of the two encoder passes without materialising copies (TwinGAN: perm = E(t), E(s), E(s), E(t)). */
int tg_upsample2x_concat_fwd(const void* x0, const void* x1, void* out, int n, int h, int w, int c0, int c1, int gsz,
                             unsigned perm, int dtype, void* stream);
/* g0[n,h,w,c0] = 2x2 sum of gout[..., :c0];  g1[n',...] = gout[..., c0:] summed over the output groups that read
 * the skip image (either may be NULL to skip) */
int tg_upsample2x_concat_bwd(const void* gout, void* g0, void* g1, int n, int h, int w, int c0, int c1, int gsz,
                             unsigned perm, int dtype, void* stream);
/* tf.nn.avg_pool 2x2 s2 VALID (nets/pggan.py:274,306,436,468): y[n,h/2,w/2,c]; scale=0.25.
 * With scale=1 it is the 2x2 sum (upsample backward). */
int tg_pool2x2_fwd(const void* x, void* y, int n, int h, int w, int c, float scale, int dtype, void* stream);
/* gx[n,h,w,c] = scale * gy[n,h/2,w/2,c] replicated 2x2 (avg-pool backward with scale .25; plain upsample with 1) */
int tg_pool2x2_bwd(const void* gy, void* gx, int n, int h, int w, int c, float scale, int dtype, void* stream);
/* out = a*x + b*y (y may be NULL).  Fade-in lerp nets/pggan.py:205,314,475; image_generation.py:1006. */
int tg_axpby(const void* x, const void* y, void* out, int64_t numel, float a, float b, int dtype, void* stream);
/* out[b,...] = x[b,...] + alpha[b] * (y[b,...] - x[b,...])   (WGAN-GP interpolates, image_generation.py:424) */
int tg_sample_lerp(const void* x, const void* y, const float* alpha, void* out, int batch, int64_t per_sample,
                   int dtype, void* stream);
/* out[b,...] = coef[b] * x[b,...] * scalar[0]   (coef fp32 [batch], scalar fp32 device pointer or NULL) */
int tg_sample_scale(const void* x, const float* coef, const float* scalar, void* out, int batch, int64_t per_sample,
                    int dtype, void* stream);
/* gdrop, mode 'prop' (libs/gdrop.py:20-36, the maybe_gdrop hook of nets/pggan.py:221-231,328-331,351-355):
 * out[n,p,c] = x[n,p,c] * (noise[n,c] * strength * sqrt(c_logical) + 1), noise fp32 [n, c] ~ N(0,1) drawn by the caller
 * (tf.random_normal there), x / out [n, hw, c] in `dtype`.  strength_dev: fp32 device scalar (the `gdrop_strength` variable
 * of image_generation.py:563-585) or NULL for the host value `strength`.  c_logical <= c: channels that count for the
 * sqrt (c is padded for the minibatch-stddev tensor).  Linear in x: the same call is its backward (and that one's). */
int tg_gdrop(const void* x, const float* noise, const float* strength_dev, float strength, int c_logical, void* out, int n,
             int64_t hw, int c, int dtype, void* stream);
/* out[i] = value * (scalar ? scalar[0] : 1)   (broadcast of a device scalar, e.g. d mean / d x) */
int tg_fill_scaled(void* out, const float* scalar, float value, int64_t numel, int dtype, void* stream);
int tg_cast(const void* src, void* dst, int64_t numel, int src_dtype, int dst_dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Minibatch stddev, nets/pggan_utils.py:353-366.  x[n, p] with p = h*w*c (4*4*C).
 * out[n,h,w,cpad]: channels [0,c) copy x, channel c = the statistic, (c, cpad) = 0 so that the
 * following 3x3 conv sees a 16-byte aligned channel count.  `groups`: the n images are `groups` consecutive
 * sub-batches (several discriminator calls batched along N), each with its OWN statistic, as in the reference
 * where every call sees one batch.  stat: fp32 [groups] or NULL.
 * ------------------------------------------------------------------------------------------- */
int tg_mbstd_fwd(const void* x, void* out, float* stat, int n, int groups, int hw, int c, int cpad, float eps, int dtype,
                 void* stream);
/* gx = gout[..., :c] + d stat/d x * sum(gout[..., c]) */
int tg_mbstd_bwd(const void* gout, const void* x, void* gx, int n, int groups, int hw, int c, int cpad, float eps,
                 int dtype, void* stream);
/* double backward: given v = grad wrt gx, returns ggout (grad wrt gout) and gx2 (grad wrt x). */
int tg_mbstd_bwd_bwd(const void* v, const void* gout, const void* x, void* ggout, void* gx2, int n, int groups, int hw,
                     int c, int cpad, float eps, int dtype, void* stream);

/* layers.fully_connected (nets/pggan_utils.py:323-327) on a network tail's [B, K] features in ONE launch each way.
 * fwd: y[m,n] fp32 = x[m,k] (`dtype`) @ w[k,n] fp32 + bias[n] (NULL: none).  bwd (first order): gx[m,k] (`dtype`) = g @ w^T,
 * gw[k,n] (+)= x^T @ g, gb[n] (+)= column sums of g; any of gx / gw / gb may be NULL; acc_w / acc_b: add into the
 * caller's buffers (gradient sinks) instead of writing.  n <= k. */
int tg_fc_fwd(const void* x, const float* w, const float* bias, float* y, int m, int n, int k, int dtype, void* stream);
int tg_fc_bwd(const void* x, const float* w, const float* g, void* gx, float* gw, float* gb, int m, int n, int k, int acc_w,
              int acc_b, int dtype, void* stream);
/* ---------------------------------------------------------------------------------------------
 * Small dense layers (layers.fully_connected, nets/pggan_utils.py:323-327; pggan.py:365-370):
 * C[m,n] = op(A)[m,k] @ op(B)[k,n] (+ bias[n]); A,B,C fp32 row-major; ta/tb transpose flags.
 * ------------------------------------------------------------------------------------------- */
int tg_small_gemm(const float* a, const float* b, const float* bias, float* c, int m, int n, int k, int ta, int tb,
                  int accumulate, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Loss reductions (twingan.py:464,502; image_generation.py:333,350,431-433).  Outputs fp32.
 * ------------------------------------------------------------------------------------------- */
/* out[0] (+)= scale * sum(x) */
int tg_sum(const void* x, float* out, int64_t numel, float scale, int accumulate, int dtype, void* stream);
/* out[0] (+)= scale * sum|a-b| */
int tg_abs_diff_sum(const void* a, const void* b, float* out, int64_t numel, float scale, int accumulate, int dtype,
                    void* stream);
/* ga = gscale[0]*scale*sign(a-b), gb = -ga  (either may be NULL) */
int tg_abs_diff_bwd(const void* a, const void* b, const float* gscale, void* ga, void* gb, int64_t numel, float scale,
                    int dtype, void* stream);
/* The loss tail of ONE batched discriminator call in one launch each way.  pred: fp32 [groups * group_size] (the
 * predictions of [real; cycle; prime] ..., image_generation.py:348-400); job j adds coef * mean_i f_mode(x_i; a, b) over
 * group `group` to terms[term] (modes as tg_pred_loss_fwd; at most 12 jobs, 8 terms; the jobs are read on the HOST and
 * travel in the kernel arguments).  bwd: gpred_i = sum over the jobs of i's group of gterms[term][0] * coef / group_size *
 * f'(x_i); gterms: HOST array of nterms device pointers to fp32 scalars (NULL: no gradient for that term). */
typedef struct TgPredJob {
  int32_t group, term, mode;
  float a, b, coef;
} TgPredJob;
int tg_pred_losses_fwd(const float* pred, int group_size, int groups, const TgPredJob* jobs, int njobs, float* terms, int nterms,
                       void* stream);
int tg_pred_losses_bwd(const float* pred, int group_size, int groups, const TgPredJob* jobs, int njobs, const float* const* gterms,
                       int nterms, float* gpred, void* stream);
/* Row blocks of a batch put together in ONE launch: for every job, dst[dst_off + i] = sum_k src[k][i], i < numel (elements of
 * `dtype`; fp32 sum rounded once; sources packed to the front of src[], a job without sources writes zeros).  The copies of
 * a concatenation along N (tf.concat(..., 0) of the reference's batched towers, twingan.py:233-288), the repeat of a batch,
 * and the backward of a tensor that is read through several row ranges.  jobs: HOST array (it travels in the kernel
 * arguments). */
#define TG_ROWS_MAX_JOBS 8
typedef struct TgRowsJob {
  const void* src[4];
  int64_t dst_off, numel;
} TgRowsJob;
int tg_rows_assemble(const TgRowsJob* jobs, int njobs, void* dst, int dtype, void* stream);
/* out[i] = lo + (hi - lo) * U[0,1), i < n (tf.random_uniform of the gradient-penalty interpolation weights and the DRAGAN
 * perturbation, image_generation.py:420-424,441-450): Philox4x32-10 keyed by (seed, state[0]).  state: TWO device words,
 * zero-initialised by the caller; the kernel advances state[0] by one per launch (so a launch captured in a hipGraph draws
 * new numbers on every replay) and uses state[1] as its rendezvous ticket. */
int tg_uniform(float* out, int64_t n, uint64_t seed, uint32_t* state, float lo, float hi, void* stream);
/* out[0] = sum of n (<= 24) device fp32 scalars, in argument order (tf.add_n over the loss collection,
 * model/model_inheritor.py); scalars: HOST array of device pointers. */
int tg_sum_scalars(const float* const* scalars, int n, float* out, void* stream);
/* Prediction losses on the fp32 [B,1] discriminator outputs (image_generation.py:331-400):
 * out[0] (+)= scale * sum_i f(x_i) with mode 0: x (WGAN means), 1: relu(a + b*x) (hinge), 2: sigmoid cross
 * entropy against label a (tf.losses.sigmoid_cross_entropy: max(x,0) - x*a + log(1+exp(-|x|))), 3: x^2 (drift
 * term :360-367).  bwd: gx_i = gscale[0] * scale * f'(x_i)  (gscale may be NULL = 1). */
int tg_pred_loss_fwd(const float* x, float* out, int n, int mode, float a, float b, float scale, int accumulate,
                     void* stream);
int tg_pred_loss_bwd(const float* x, const float* gscale, float* gx, int n, int mode, float a, float b, float scale,
                     void* stream);
/* out[0] = variance over every element, from sum[0] = sum(x) and sample_sumsq[batch] (DRAGAN's "std",
 * image_generation.py:445, which is tf.nn.moments(...)[1]) */
int tg_var_from_sums(const float* sum, const float* sample_sumsq, float* out, int batch, int64_t numel, void* stream);
/* out[b] = sum over the sample of x^2 */
int tg_sample_sumsq(const void* x, float* out, int batch, int64_t per_sample, int dtype, void* stream);
/* WGAN-GP scalar tail: loss[0] = lambda*mean_b (sqrt(ss[b])-1)^2 ; coef[b] = lambda*2*(sqrt(ss)-1)/(sqrt(ss)*batch) */
int tg_gp_penalty(const float* sumsq, float* loss, float* coef, int batch, float lambda, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Adam, TF-1.x form (model/model_inheritor.py:537-542): theta -= lr_t * m / (sqrt(v) + eps) with
 * lr_t = lr*sqrt(1-b2^t)/(1-b1^t) computed by the caller.  Flat fp32 buffers of `numel`
 * elements; grad_scale multiplies g first (1/loss_scale).  theta_bf16 (may be NULL) receives the
 * rounded copy of the updated parameters (cast-on-read shadow, deployment/model_deploy.py:146-183).
 * lr_t_dev (may be NULL): device fp32 [1] that overrides lr_t -- lets a captured hipGraph replay the
 * step with a fresh bias-corrected rate.  tg_adam_tick advances the shared optimiser step counter on
 * the device (one counter for G and D applies, image_generation.py:554-561) and writes that rate.
 * ------------------------------------------------------------------------------------------- */
int tg_adam_step(float* theta, const float* grad, float* m, float* v, void* theta_bf16, int64_t numel, float lr_t,
                 const float* lr_t_dev, float beta1, float beta2, float eps, float grad_scale, void* stream);
int tg_adam_tick(int64_t* step_dev, float* lr_t_dev, float lr, float beta1, float beta2, void* stream);

/* -------------------------------------------------------------------------------------------
 * SAGAN self-attention -- replaces tf.matmul (x2), tf.nn.softmax, tf.nn.tanh and gamma * o + layer of
 * libs/self_attention.py:57-69 (called from nets/pggan_utils.py:301-308 under --do_self_attention).  The layer is
 * composed on the host from these entry points; each is closed under differentiation (the backward of a product is two
 * products, the softmax backward has its own gradient kernel), so the layer is differentiable twice on them -- the
 * discriminators sit under the WGAN-GP penalty (image_generation.py:414-439).
 *   tg_batched_gemm: C[i] = alpha * op(A[i]) op(B[i]) (+ C[i]), i < batch; op(A) is m x k stored [m][lda] (ta = 0) or
 *     [k][lda] (ta = 1), op(B) is k x n stored [k][ldb] (tb = 0) or [n][ldb] (tb = 1); strides in elements.
 *     dtype TG_BF16: MFMA kernel (C bf16, or fp32 with c_is_f32); TG_F32: exact fp32.
 *   tg_softmax_rows_fwd: p = softmax(s) over the last axis of [rows, cols];  _bwd: ds = p * (dp - sum(dp * p));
 *     _bwd_bwd: gp = d (sum v * ds) / d p = v * (dp - sum(dp p)) - dp * sum(v p)   (d / d dp is _bwd(p, v)).
 *   tg_tanh_fwd / _bwd (gx = g (1 - y^2)), tg_mul3 (out = scale a b c; c may be NULL), tg_scale_dev (out = x * s[0],
 *     s on the device: sa_gamma), tg_dot (out[0] = sum a b; ws >= 1024 floats).
 * ------------------------------------------------------------------------------------------- */
int tg_batched_gemm(const void* a, const void* b, void* c, int batch, int m, int n, int k, int ta, int tb, int lda, int ldb,
                    int ldc, int64_t stride_a, int64_t stride_b, int64_t stride_c, float alpha, int accumulate, int dtype,
                    int c_is_f32, void* stream);
int tg_softmax_rows_fwd(const void* s, void* p, int64_t rows, int cols, int dtype, void* stream);
int tg_softmax_rows_bwd(const void* p, const void* dp, void* ds, int64_t rows, int cols, int dtype, void* stream);
int tg_softmax_rows_bwd_bwd(const void* p, const void* dp, const void* v, void* gp, int64_t rows, int cols, int dtype,
                            void* stream);
int tg_tanh_fwd(const void* x, void* y, int64_t numel, int dtype, void* stream);
int tg_tanh_bwd(const void* g, const void* y, void* gx, int64_t numel, int dtype, void* stream);
int tg_mul3(const void* a, const void* b, const void* c, void* out, float scale, int64_t numel, int dtype, void* stream);
int tg_scale_dev(const void* x, const float* scalar, void* out, int64_t numel, int dtype, void* stream);
int tg_dot(const void* a, const void* b, float* out, float* ws, int64_t numel, int dtype, void* stream);

/* -------------------------------------------------------------------------------------------
 * Encoder-distillation loss -- replaces tf.nn.l2_normalize x2 + tf.losses.cosine_distance (twingan.py:507-521):
 *   out[0] = (weight / batch) * sum_b (1 - l2n(expected_b) . l2n(embedding_b)),  fp32 [batch, dim] operands.
 *   bwd: g_embedding = d out / d embedding * gscale[0]  (expected is dataset input: no gradient).
 * ------------------------------------------------------------------------------------------- */
int tg_cosine_distance_fwd(const float* expected, const float* embedding, float* out, int batch, int dim, float weight,
                           void* stream);
int tg_cosine_distance_bwd(const float* expected, const float* embedding, const float* gscale, float* g_embedding, int batch,
                           int dim, float weight, void* stream);

/* -------------------------------------------------------------------------------------------
 * Spectral normalisation of a conv kernel -- replaces libs/sn.py:38-101 (spectral_normed_weight: the tf.matmul /
 * tf.nn.l2_normalize chain behind libs.sn.convolution, nets/pggan_utils.py:316-320, --spectral_norm).
 * w: fp32 [k_rows = kh*kw*cin, cout] (the HWIO kernel flattened), u: fp32 [cout] persistent power-iteration vector.
 *   fwd: v = l2n(u W^T) [k_rows], u_new = l2n(v W) [cout], sigma = v W u_new^T, w_bar = W / sigma;
 *        stats = {sigma, |u W^T|} (fp32 [2], kept for the backward).  The caller assigns u <- u_new (libs/sn.py:84-86).
 *   bwd: gw (+)= d L / d W for g_wbar = d L / d w_bar, the gradient flowing through sigma, v and u_new as in the
 *        reference (no stop_gradient).
 * ws: tg_spectral_norm_workspace(k_rows, cout) bytes of scratch.  cout <= 1024.
 * ------------------------------------------------------------------------------------------- */
size_t tg_spectral_norm_workspace(int k_rows, int cout);
int tg_spectral_norm_fwd(const float* w, const float* u, float* w_bar, float* u_new, float* v, float* stats, int k_rows,
                         int cout, void* ws, size_t ws_bytes, void* stream);
int tg_spectral_norm_bwd(const float* g_wbar, const float* w, const float* u, const float* u_new, const float* v,
                         const float* stats, float* gw, int accumulate, int k_rows, int cout, void* ws, size_t ws_bytes,
                         void* stream);
/* The power iteration of MANY kernels in three launches (the trainer's start-of-run pass under --spectral_norm: 60 matrices
 * in BASELINE configs[4]).  A job table in device memory: tg_sn_table_bytes(njobs) bytes, filled on the host job by job with
 * tg_sn_table_fill (arguments as tg_spectral_norm_fwd; `totals` = three running block counts, zeroed before job 0) and copied
 * to the device by the caller; tg_spectral_norm_fwd_multi(table, njobs, totals[0], totals[1], totals[2]) then computes every
 * job's w_bar / u_new / v / stats exactly as tg_spectral_norm_fwd would (same kernels' bodies, same summation order). */
size_t tg_sn_table_bytes(int njobs);
int tg_sn_table_fill(int j, const float* w, const float* u, float* w_bar, float* u_new, float* v, float* stats, void* ws,
                     size_t ws_bytes, int k_rows, int cout, void* host_table, int32_t* totals);
int tg_spectral_norm_fwd_multi(const void* table, int njobs, int row_blocks, int col_blocks, int fin_blocks, void* stream);
/* u <- u_new of every job of the table, one launch: the assign of libs/sn.py:84-86 at the end of a run (the caller's u
 * buffers are the table's `u` entries, read by tg_spectral_norm_fwd_multi of the NEXT run). */
int tg_sn_assign_u(const void* table, int njobs, void* stream);

/* SAGAN self-attention (libs/self_attention.py:24-70: s = tf.matmul(f, g, transpose_b=True) over the h*w positions,
 * beta = tf.nn.softmax(s), o = tf.matmul(beta, h)) without materialising the [len x len] map: q = f [n, len, dk],
 * k = g [n, len, dk], v = h [n, len, dv], o [n, len, dv]; 16-bit storage, fp32 softmax statistics and accumulation.
 * The kernels fetch their per-tile MFMA operands from fragment-ordered copies of v / d_o (and transposes of q / k) that
 * the entry points build themselves in `workspace` (tg_flash_attention_workspace_bytes(n, len, dk, dv, pass) bytes -- pass 0: fwd, 1: bwd, 2: bwd_bwd -- of
 * device memory, 256-byte aligned, contents undefined afterwards; 0 = unsupported shape).
 * tg_flash_attention_supported: len % 128 == 0, dk in {8, 16}, dv in {64, 128, 256}.
 * fwd: writes o and lse [n, len] fp32 (log-sum-exp of every query's scores, saved for the backward).
 * bwd (first order; the gradient-penalty double backward keeps the tg_batched_gemm / tg_softmax_rows composition):
 * given d_o, o, lse writes dq, dk_out [n, len, dk] and dv_out [n, len, dv]; dv in {64, 128}.
 * bwd_bwd (the gradient penalty differentiates the first-order backward, image_generation.py:414-439): given the
 * cotangents a_q, a_k [n, len, dk], a_v [n, len, dv] of (dq, dk_out, dv_out) writes the gradients of their sum of products
 * with respect to q, k, v and d_o (adj_q, adj_k [n, len, dk]; adj_v, adj_do [n, len, dv]); dv in {64, 128};
 * workspace: tg_flash_attention_workspace_bytes(..., 2).
 * tg_transpose16: [batch, rows, cols] -> [batch, cols, rows] of 16-bit elements. */
int tg_transpose16(const void* src, void* dst, int batch, int rows, int cols, void* stream);
int tg_flash_attention_supported(int len, int dk, int dv);
int64_t tg_flash_attention_workspace_bytes(int n, int len, int dk, int dv, int backward);
int tg_flash_attention_fwd(const void* q, const void* k, const void* v, void* o, float* lse, void* workspace, int n, int len,
                           int dk, int dv, int dtype, void* stream);
int tg_flash_attention_bwd(const void* q, const void* k, const void* v, const void* d_o, const void* o, const float* lse,
                           void* workspace, void* dq, void* dk_out, void* dv_out, int n, int len, int dk, int dv, int dtype,
                           void* stream);
int tg_flash_attention_bwd_bwd(const void* q, const void* k, const void* v, const void* d_o, const void* o, const float* lse,
                               const void* a_q, const void* a_k, const void* a_v, void* workspace, void* adj_q, void* adj_k,
                               void* adj_v, void* adj_do, int n, int len, int dk, int dv, int dtype, void* stream);

/* Gradient all-reduce for callers without torch.distributed (the reference sums its clones' gradients in one process,
 * deployment/model_deploy.py:473-503; with one process per GPU that sum is a sum all-reduce over xGMI): a thin wrapper
 * over RCCL, bound lazily (dlopen of $TG_RCCL_PATH, librccl.so.1 or librccl.so at the first call; the library does not
 * link RCCL).  Rank 0 obtains an id (tg_comm_unique_id_bytes() bytes) and distributes it out of band; every rank calls
 * tg_comm_init with its current HIP device set; tg_allreduce sums `count` elements of dtype TG_F32 / TG_BF16 / TG_F16 IN
 * PLACE, asynchronously on `stream`.  An RCCL failure returns TG_ECOMM. */
int tg_comm_unique_id_bytes(void);
int tg_comm_unique_id(void* id);
int tg_comm_init(const void* id, int nranks, int rank, void** comm);
int tg_allreduce(void* comm, void* buf, int64_t count, int dtype, void* stream);
int tg_comm_destroy(void* comm);

/* Training-image preprocessing (preprocessing/danbooru_preprocessing.py:115-230 preprocess_image with the TwinGAN
 * trainer's defaults, model/model_inheritor.py:403-457; resize_image of preprocessing/preprocessing_util.py:97-146): n
 * decoded uint8 RGB images of arbitrary size -> out [n, hw, hw, 3] (dtype) in [0, 1].  packed: the images' bytes, image i
 * at packed + offsets[i] as [h][w][3]; rect: int32 [n][6] = image h, w and the source rectangle (y0, x0, sh, sw) in
 * image coordinates that is resized to hw x hw -- PAD: (-(size-h)/2, -(size-w)/2, size, size) with size = max(h, w) and
 * zeros outside the image, CROP: ((h-size)/2, (w-size)/2, size, size) with size = min(h, w), RESHAPE: (0, 0, h, w);
 * bilinear as TF 1.x (align_corners=False, no half-pixel centres).  aug: fp32 [n][4] = flip left-right (0/1), colour
 * order (0: brightness then saturation, 1: saturation then brightness), brightness delta, saturation factor -- the
 * random draws of random_flip_left_right / distort_color(fast_mode) are the caller's; (0, 0, 0, 1) = evaluation. */
int tg_preprocess_images(const void* packed, const int64_t* offsets, const int* rect, const float* aug, void* out, int n,
                         int hw, int dtype, void* stream);
/* The same with --do_random_cropping (model/model_inheritor.py:225,449-454; docs/training.md:22-23 trains with it;
 * danbooru_preprocessing.py:187-201, preprocessing_util.random_crop_image :312-331) and --color_space
 * (model_inheritor.py:240,414; danbooru_preprocessing.py:208-225).  crop (NULL: none): int32 [n][4] = (cy, cx, ch, cw), the
 * rectangle tf.random_crop cuts out of the INTERMEDIATE image -- the source rectangle resized to mid x mid, mid =
 * int(hw / random_cropping_ratio) -- which a second bilinear resize brings to hw x hw; 0 <= cy, cy + ch <= mid (same for x);
 * the draws (ch, cw = int32(mid * U[ratio, 1)), offsets uniform over the valid range) are the caller's.  color_space:
 * 0 rgb, 1 yiq (preprocessing_util.rgb_to_yiq :154-160, applied after the distortion and clip), 2 bgr (channel reverse),
 * 3 gray (the colour distortion is skipped; the image keeps its three channels, as in the reference). */
int tg_preprocess_images_crop(const void* packed, const int64_t* offsets, const int* rect, const int* crop, const float* aug,
                              void* out, int n, int hw, int mid, int color_space, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TWINGAN_HIP_H_ */
