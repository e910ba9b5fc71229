// conv_img: 3x3 SAME stride-1 convolution over the 8x8 and 4x4 maps (the deep stages of E / G / D: 256 -> 256 and
// 512 -> 256 channels, nets/pggan.py:148-166,289-315,450-476), forward and backward-data (rotated pack), bf16 / f16
// NHWC in and out, fp32 accumulate on v_mfma_f32_32x32x16.
//
// conv_small reads BOTH operands of every MFMA straight from L2 in fragment layout: a 64-pixel x 32-channel workgroup
// fetches 9 taps x 64 pixels x cin of activations (each input element nine times: 295 KB at cin 256) next to its 147 KB
// weight slice -- 110 MB of L2 -> CU traffic for a 2.4 GFLOP launch (8x8, 256 -> 256, n 32: 19 us, 0.05 of the MFMA peak).
// Here a workgroup owns whole images: it stages their zero-padded halo ONCE in LDS (coalesced 16-byte loads, 51 KB per
// 8x8 image at cin 256) and reads every tap's pixel fragments from there; only the weight fragments still come from L2,
// three 9-tap stages ahead of their MFMAs (the registers the pixel fragments no longer need).
//   workgroup = 32*MT output pixels (MT = 2: one 8x8 image or four 4x4 images) x 32 output channels, 4 waves that SPLIT K
//   (wave w: 16-channel chunks w, w+4, ...), one LDS reduction at the end, conv_small's epilogue.
// Reference call sites replaced: tf.contrib.layers.conv2d at nets/pggan_utils.py:316-320 for those stages and its
// Conv2DBackpropInput gradient.
#include "tg_common.h"

namespace {

struct ImgGeom {
  int n, cin, cout;            // images, input / output channels (cin % 32 == 0, cout % 32 == 0)
  int cin_pad;                 // channels per tap in the weight pack
  int epilogue;
  float alpha;
  unsigned w_bytes;
  // STATS kernels: sums of the (16-bit-rounded) outputs and of their squares per image and output channel, written as the
  // ONE statistics chunk of the image -- stats[img][0][2][cout], the layout of conv_tile's STATS epilogue with
  // stat_chunks = 1 -- so the instance norm after an 8x8 conv needs no pass over the tensor (in_stats_partial, norm.hip)
  float* stats;
  // masked backward-data: the output is multiplied by the LeakyReLU derivative of `mask` (the forward input of the layer
  // this backward-data belongs to = the producer's activation output, same shape as y): mask > 0 ? 1 : alpha.  NULL: plain.
  // (conv_tile's mask epilogue for the 8x8 / 4x4 maps: their tg_lrelu_bwd launches -- one per block of every discriminator
  // pass -- are gone)
  const bf16* mask;
  // weight-set groups (TgConvDesc::groups): npg > 0 = images per group; image i uses weight set i / npg, whose pack starts
  // wgs_bytes after the previous one (bias row: cout floats).  8x8 maps only (one image per workgroup).
  int npg;
  unsigned wgs_bytes;
};

constexpr unsigned IOOB = 0x80000000u;
typedef __attribute__((ext_vector_type(4))) unsigned iu32x4;
extern __shared__ __attribute__((aligned(16))) unsigned char img_smem[];

__device__ __forceinline__ __amdgpu_buffer_rsrc_t i_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// HW: map size (8 or 4).  MT: 32-pixel column blocks per workgroup (images per workgroup = 32 * MT / HW^2).
template <int HW, int MT, bool F16 = false, bool STATS = false>
__global__ __launch_bounds__(256, 2) void conv_img_kernel(const bf16* __restrict__ x, const bf16* __restrict__ wp,
                                                       const float* __restrict__ bias0, bf16* __restrict__ y,
                                                       const ImgGeom g) {
  constexpr int PPI = HW * HW, HD = HW + 2, IMGS = 32 * MT / PPI, NT = 9;
  static_assert(IMGS >= 1 && IMGS * PPI == 32 * MT, "a workgroup holds whole images");
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, kgrp = lane >> 5;
  const int img0 = blockIdx.x * IMGS;               // first image of this workgroup
  const int n0 = blockIdx.y * 32;                   // first output channel
  const int ps = g.cin * 2 + 16;                    // LDS bytes per halo pixel (16-byte pad: rows of pixels spread over the banks)
  const int vpp = g.cin >> 3;                       // 16-byte vectors per pixel

  // ---- weight fragments of the first stages go out before anything else: they do not depend on the image
  const int wset = g.npg ? (img0 >= g.npg) + (img0 >= 2 * g.npg) + (img0 >= 3 * g.npg) : 0;      // uniform: IMGS == 1 when grouped
  const float* bias = bias0 + wset * g.cout;      // only read under TG_EPI_BIAS
  const __amdgpu_buffer_rsrc_t rw = i_rsrc(reinterpret_cast<const unsigned char*>(wp) + (size_t)wset * g.wgs_bytes, g.w_bytes);
  // fragment-ordered pack (conv_mfma.hip pack_coords): [row block][K chunk][tap][32 rows][16 channels] -- the wave's fragment
  // of (tap, chunk) is one contiguous KB in lane order, a chunk's nine taps nine consecutive KB
  const int nchunks = g.cin >> 4;                   // 16-channel K chunks; wave w takes w, w + 4, ...
  const unsigned wblk = (unsigned)((n0 >> 5) * (g.cin_pad >> 4)) * (NT * 1024u);      // this workgroup's row block
  const unsigned woff = wblk + (unsigned)((l31 * 16 + kgrp * 8) * 2);                   // + (chunk * 9 + tap) * 1024
  struct WStage {
    bf16x8 w[NT];
  };
  auto load_w = [&](WStage& st, int ck) __attribute__((always_inline)) {
    const bool in = ck < nchunks;
#pragma unroll
    for (int t = 0; t < NT; ++t)
      st.w[t] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                               rw, in ? woff + (unsigned)((ck * NT + t) * 1024) : IOOB, 0, 0));
  };
  WStage wa, wb, wc;
  load_w(wa, wid);
  load_w(wb, wid + 4);
  load_w(wc, wid + 8);

  // ---- stage the images' zero-padded halos: IMGS * HD * HD pixels x cin channels, 16 bytes per thread and trip
  {
    const size_t img_elems = (size_t)PPI * g.cin;
    const int nimg = min(IMGS, g.n - img0);
    const __amdgpu_buffer_rsrc_t rx = i_rsrc(x + (size_t)img0 * img_elems, (unsigned)(nimg * img_elems * 2));
    const int total = IMGS * HD * HD * vpp;
    const int vshift = 31 - __builtin_clz(vpp);      // vpp = cin / 8 is a power of two for the layers taken (checked by the host)
    constexpr int UL = 13;                            // loads in flight per thread: an 8x8 image at 256 channels is ONE trip (12.5 vectors per thread), at 512 two
    for (int base = tid; base < total; base += 256 * UL) {
      iu32x4 val[UL];
#pragma unroll
      for (int u = 0; u < UL; ++u) {
        const int v = base + u * 256;
        const int px = v >> vshift, part = v & (vpp - 1);
        const int il = px / (HD * HD), rem = px - il * (HD * HD);
        const int hy = rem / HD - 1, hx = rem % HD - 1;
        const bool ok = v < total && il < nimg && hy >= 0 && hy < HW && hx >= 0 && hx < HW;
        val[u] = __builtin_amdgcn_raw_buffer_load_b128(
            rx, ok ? (unsigned)((((il * HW + hy) * HW + hx) * g.cin + part * 8) * 2) : IOOB, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < UL; ++u) {
        const int v = base + u * 256;
        if (v < total) *reinterpret_cast<iu32x4*>(img_smem + (v >> vshift) * ps + (v & (vpp - 1)) * 16) = val[u];
      }
    }
  }
  __syncthreads();

  // ---- this lane's pixel of each column block: halo slot of its (ky, kx) = (0, 0) tap
  int pslot[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int pidx = m * 32 + l31;
    const int il = pidx / PPI, rem = pidx - il * PPI;
    const int py = rem / HW, pxx = rem - py * HW;
    pslot[m] = ((il * HD + py) * HD + pxx) * ps + kgrp * 16;
  }

  f32x16 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[m][j] = 0.f;

  // one K chunk: 9 taps x MT column blocks; the pixel fragments of a tap are 16 bytes per lane from LDS
  auto compute = [&](const WStage& st, int ck) __attribute__((always_inline)) {
    const int coff = ck * 32;      // byte offset of the chunk's first channel inside a pixel
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int toff = ((t / 3) * HD + (t % 3)) * ps + coff;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(img_smem + pslot[m] + toff);
        acc[m] = mfma_32x32x16<F16>(st.w[t], xf, acc[m]);
      }
    }
  };
  // three weight stages in rotation: stage i + 3 is requested before the MFMAs of stage i
  for (int ck = wid; ck < nchunks; ck += 12) {
    compute(wa, ck);
    load_w(wa, ck + 12);
    if (ck + 4 >= nchunks) break;
    compute(wb, ck + 4);
    load_w(wb, ck + 16);
    if (ck + 8 >= nchunks) break;
    compute(wc, ck + 8);
    load_w(wc, ck + 20);
  }

  // ---- sum the 4 K slices through LDS (the image is no longer needed), one column block at a time; conv_small's epilogue
  __syncthreads();
  float* red = reinterpret_cast<float*>(img_smem);      // [4][16][64] floats = 16 KiB
  const size_t out_elems = (size_t)PPI * g.cout;
  const int nimg = min(IMGS, g.n - img0);
  const __amdgpu_buffer_rsrc_t rbias = i_rsrc(bias, (g.epilogue & TG_EPI_BIAS) ? (unsigned)(g.cout * 4) : 0u);
  const __amdgpu_buffer_rsrc_t ry = i_rsrc(y + (size_t)img0 * out_elems, (unsigned)(nimg * out_elems * 2));
  const __amdgpu_buffer_rsrc_t rmask = i_rsrc(g.mask ? g.mask + (size_t)img0 * out_elems : y, g.mask ? (unsigned)(nimg * out_elems * 2) : 0u);
  const int q = wid;      // wave w finishes register quads q = w: channels 8w + 4 kgrp .. + 3 of the 32-block
  const f32x4 bq = __builtin_bit_cast(
      f32x4, __builtin_amdgcn_raw_buffer_load_b128(rbias, (unsigned)((n0 + q * 8 + kgrp * 4) * 4), 0, 0));
  float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};      // STATS: this lane's pixels, channels 8q + 4 kgrp + j
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    if (m) __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wid * 16 + r) * 64 + lane] = acc[m][r];
    __syncthreads();
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = q * 4 + j;
      v[j] = red[(0 * 16 + r) * 64 + lane] + red[(1 * 16 + r) * 64 + lane] + red[(2 * 16 + r) * 64 + lane] +
             red[(3 * 16 + r) * 64 + lane] + bq[j];
      if (g.epilogue & TG_EPI_LRELU) v[j] = lrelu_f(v[j], g.alpha);
    }
    if (g.mask) {      // uniform: a positive bf16 / f16 is a positive int16 pattern
      typedef __attribute__((ext_vector_type(2))) unsigned iu32x2;
      const iu32x2 z = __builtin_bit_cast(iu32x2, __builtin_amdgcn_raw_buffer_load_b64(
          rmask, (unsigned)(((m * 32 + l31) * g.cout + n0 + q * 8 + kgrp * 4) * 2), 0, 0));
      v[0] *= (short)(z[0] & 0xffffu) > 0 ? 1.f : g.alpha;
      v[1] *= (short)(z[0] >> 16) > 0 ? 1.f : g.alpha;
      v[2] *= (short)(z[1] & 0xffffu) > 0 ? 1.f : g.alpha;
      v[3] *= (short)(z[1] >> 16) > 0 ? 1.f : g.alpha;
    }
    const unsigned p0 = pack16x2<F16>(v[0], v[1]), p1 = pack16x2<F16>(v[2], v[3]);
    if constexpr (STATS) {      // of the values as stored
      const float r4[4] = {unpack16_lo<F16>(p0), unpack16_hi<F16>(p0), unpack16_lo<F16>(p1), unpack16_hi<F16>(p1)};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ssum[j] += r4[j];
        ssq[j] = fmaf(r4[j], r4[j], ssq[j]);
      }
    }
    // low lanes hold channels 8q..8q+3, high lanes 8q+4..8q+7 of the same pixel: give the low lane all 8
    auto s0 = __builtin_amdgcn_permlane32_swap(p0, p0, false, false);
    auto s1 = __builtin_amdgcn_permlane32_swap(p1, p1, false, false);
    iu32x4 o;
    o[0] = p0; o[1] = p1; o[2] = s0[1]; o[3] = s1[1];
    const int ch0 = n0 + q * 8;
    const int p = m * 32 + l31;      // pixel inside the workgroup's images (images are contiguous in NHWC)
    const bool ok = kgrp == 0 && ch0 + 8 <= g.cout;      // pixels of images past the batch fall outside `ry`
    __builtin_amdgcn_raw_buffer_store_b128(o, ry, ok ? (unsigned)((p * g.cout + ch0) * 2) : IOOB, 0, TG_STORE_AUX);
  }
  if constexpr (STATS) {
    // the workgroup holds ONE whole image: a butterfly over the 32 pixel lanes of each half-wave (fixed order) leaves the
    // image's sums of 4 channels in every lane; lane 0 of each half writes them
    static_assert(IMGS == 1, "statistics epilogue: one image per workgroup");
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) {
        ssum[j] += __shfl_xor(ssum[j], o, 64);
        ssq[j] += __shfl_xor(ssq[j], o, 64);
      }
    }
    if (l31 == 0 && img0 < g.n) {
      float* out = g.stats + (size_t)img0 * 2 * g.cout + n0 + q * 8 + kgrp * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        out[j] = ssum[j];
        out[g.cout + j] = ssq[j];
      }
    }
  }
}

template <int HW, int MT, bool STATS = false>
int launch_img(const ImgGeom& g, const void* x, const void* wp, const float* bias, void* y, hipStream_t s) {
  constexpr int IMGS = 32 * MT / (HW * HW), HD = HW + 2;
  const size_t lds_img = (size_t)IMGS * HD * HD * (g.cin * 2 + 16);
  const size_t lds = lds_img > 16384 ? lds_img : 16384;      // the reduction scratch reuses the image's bytes
  TG_CHECK(lds <= 160 * 1024, TG_ENOSUP, "conv_img: LDS %zu too large", lds);
  const dim3 grid((g.n + IMGS - 1) / IMGS, g.cout / 32);
  const bool f16 = tg_elem_f16();
  auto k0 = conv_img_kernel<HW, MT, false, STATS>;
  auto k1 = conv_img_kernel<HW, MT, true, STATS>;
  if (lds > 64 * 1024) {
    static unsigned long long raised = 0;      // per (HW, MT, STATS), one bit per device
    if (tg_first_on_device(&raised)) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(k0), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
        tg_set_error("conv_img: cannot raise dynamic LDS to %zu", lds);
        return TG_ELAUNCH;
      }
    }
  }
  if (STATS) tg_note_kernel(f16 ? "conv_img_kernel<%d,%d,f16,stats>" : "conv_img_kernel<%d,%d,stats>", HW, MT);
  else if (g.npg) tg_note_kernel(f16 ? "conv_img_kernel<%d,%d,f16,sets>" : "conv_img_kernel<%d,%d,sets>", HW, MT);
  else tg_note_kernel(f16 ? "conv_img_kernel<%d,%d,f16>" : "conv_img_kernel<%d,%d>", HW, MT);
  if (f16) hipLaunchKernelGGL(k1, grid, dim3(256), lds, s, (const bf16*)x, (const bf16*)wp, bias, (bf16*)y, g);
  else hipLaunchKernelGGL(k0, grid, dim3(256), lds, s, (const bf16*)x, (const bf16*)wp, bias, (bf16*)y, g);
  TG_LAUNCH_CHECK("conv_img");
  return TG_OK;
}

}  // namespace

// 3x3 SAME on square 8x8 maps with channel counts the fragment loads take whole (cin % 32: a 16-channel chunk per wave
// and step).  Measured against conv_small on one box, us conv_small / conv_img (gpurun_out r3j, forward = backward-data):
// 256 -> 256: n 16 10.5 / 10.2, n 32 15.3 / 10.8, n 64 24.1 / 16.5; 512 -> 256: n 16 18.4 / 16.8 (dgrad 15.3 / 10.9), n 32
// 26.6 / 18.0, n 64 42.5 / 34.2 (dgrad 46.8 / 30.2); bench step 857.7 -> 870.5 images/s (+1.5 %).  The 4x4 maps (kernel
// instantiated: four images per workgroup) measured SLOWER, 9.7-10.0 / 10.6-10.9 us at every n -- 16 pixels per image
// leave the staging nothing to amortise -- and stay on conv_small, like the 264-channel minibatch-stddev layer.
bool tg_conv_img_supported(int n, int hin, int win, int cin, int hout, int wout, int cout, int k, int pad_t, int pad_l) {
  if (k != 3 || pad_t != 1 || pad_l != 1 || hin != hout || win != wout || hin != win) return false;
  if (hin != 8) return false;      // the 4x4 maps measured slower than conv_small (see above)
  if (cin % 32 != 0 || cout % 32 != 0 || cin > 1024 || n < 1 || (cin & (cin - 1)) != 0) return false;      // cin / 8 a power of two
  if (hin == 8 && (size_t)100 * (cin * 2 + 16) > 160 * 1024) return false;
  return true;
}

// the statistics epilogue exists for the 8x8 maps (one image per workgroup), plain epilogue
bool tg_conv_img_stats_supported(int n, int hin, int win, int cin, int hout, int wout, int cout, int k, int pad_t, int pad_l) {
  return hin == 8 &&
         tg_conv_img_supported(n, hin, win, cin, hout, wout, cout, k, pad_t, pad_l);
}

int tg_conv_img_run(int n, int hw, int cin, int cout, int epilogue, float alpha, const void* x, const void* wp,
                    const float* bias, void* y, hipStream_t s, float* stats, const void* mask, int groups, size_t wset_elems) {
  ImgGeom g;
  TG_CHECK(groups <= 1 || (hw == 8 && !stats), TG_ENOSUP, "conv_img: weight-set groups take the 8x8 maps (one image per workgroup)");
  g.npg = groups > 1 ? n / groups : 0;
  g.wgs_bytes = groups > 1 ? (unsigned)(wset_elems * 2) : 0u;
  g.stats = stats;
  g.mask = (const bf16*)mask;
  TG_CHECK(!(mask && (stats || epilogue)), TG_ENOSUP, "conv_img: the mask epilogue comes with the plain epilogue only");
  g.n = n; g.cin = cin; g.cout = cout;
  g.cin_pad = (cin + 15) / 16 * 16;
  g.epilogue = epilogue;
  g.alpha = alpha;
  const size_t rows_pad = (size_t)(cout + 63) / 64 * 64;
  const size_t wb = rows_pad * 9 * g.cin_pad * 2;
  TG_CHECK(wb < 0x7fffffffull && (size_t)n * hw * hw * (cin > cout ? cin : cout) * 2 < 0x7fffffffull, TG_ENOSUP,
           "conv_img: tensor too large");
  g.w_bytes = (unsigned)wb;
  if (stats) {
    TG_CHECK(hw == 8 && epilogue == 0, TG_ENOSUP, "conv_img: the statistics epilogue takes 8x8 maps and the plain epilogue");
    return launch_img<8, 2, true>(g, x, wp, bias, y, s);
  }
  if (hw == 8) return launch_img<8, 2>(g, x, wp, bias, y, s);
  return launch_img<4, 2>(g, x, wp, bias, y, s);
}
