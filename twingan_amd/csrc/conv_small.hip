// conv_small: stride-1 convolution (any k <= 3, any zero padding; also the "k x k VALID on a k x k input"
// layers rewritten as dense 1x1) for SMALL pixel counts -- the 4x4 / 8x8 stages and the discriminator
// tail, where N*H*W is 16..1024 but Cin*k*k is 2304..4608.  bf16 NHWC in/out, fp32 accumulate on
// v_mfma_f32_32x32x16_bf16.  Forward and backward-data (rotated pack).
//
// These layers are latency-bound, not bandwidth- or MFMA-bound: the first MFMA kernel gave them 8-32
// workgroups that each walked the whole K dimension (25-70 us).  Here
//   - a workgroup owns a 32-pixel x 32-channel output tile (grid = pixels/32 x cout/32),
//   - its 4 waves SPLIT K: wave w takes the 16-channel chunks w, w+4, w+8, ... of every tap,
//   - operands go straight from L2 to registers in MFMA fragment layout (16 B per lane; activations and
//     weights of these layers are L2-resident), no LDS staging, no barrier in the main loop,
//   - out-of-image taps are buffer loads with an out-of-range offset (return 0),
//   - the 4 partial accumulators are summed through LDS once, then bias + LeakyReLU + 16-byte stores.
//
// Reference call sites replaced: the 4x4 / 8x8 convs of nets/pggan.py:148-166,289-315,318-335,450-476
// (tf.contrib.layers.conv2d, nets/pggan_utils.py:316-320) and their Conv2DBackpropInput gradients.
#include "tg_common.h"

namespace {

struct SmallGeom {
  int n, hin, win, cin, hout, wout, cout;
  int cin_pad, kh, kw, pad_t, pad_l;
  int npix;                    // n * hout * wout
  int nchunks;                 // cin_pad / 16
  int epilogue;
  float alpha;
  unsigned x_bytes, w_bytes, y_bytes;
  // STATS kernels (4x4 maps): sums of the (16-bit-rounded) outputs and of their squares per image and output channel as
  // the image's ONE statistics chunk, stats[img][0][2][cout] (the layout of conv_tile's STATS epilogue, stat_chunks = 1)
  float* stats;
  // masked backward-data: y *= (mask > 0 ? 1 : alpha), mask = the forward input of the layer (same shape as y); NULL: plain
  const bf16* mask;
  // weight-set groups (TgConvDesc::groups): bpg > 0 = workgroups (along x) per group; group i owns the pixels
  // [i * gpix, (i + 1) * gpix) and uses weight set i, whose pack starts wgs_bytes after the previous one (bias row: cout
  // floats).  A workgroup never straddles two groups (the MFMA's weight operand is shared by its pixel columns).
  int bpg, gpix;
  unsigned wgs_bytes;
};

constexpr unsigned SOOB = 0x80000000u;
typedef __attribute__((ext_vector_type(4))) unsigned su32x4;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t s_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ bf16x8 s_load16(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}

template <int NT, int MT, int U, bool F16 = false, bool STATS = false>      // taps (1 or 9); 32-pixel column blocks per workgroup; chunks per load group
__global__ __launch_bounds__(256) void conv_small_kernel(const bf16* __restrict__ x, const bf16* __restrict__ wp,
                                                         const float* __restrict__ bias0, bf16* __restrict__ y,
                                                         const SmallGeom g) {
  __shared__ float red[4][16][64];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, kgrp = lane >> 5;
  int bx = blockIdx.x, wset = 0;
  if (g.bpg) {      // uniform
    wset = bx / g.bpg;
    bx -= wset * g.bpg;
  }
  const int npix = g.bpg ? (wset + 1) * g.gpix : g.npix;      // end of this workgroup's pixel range
  const int pbase = wset * g.gpix + bx * 32 * MT + l31;      // this lane's output pixel of column block 0 (B operand column)
  const int n0 = blockIdx.y * 32;                   // first output channel of the tile
  const float* bias = bias0 + wset * g.cout;        // only read under TG_EPI_BIAS

  const __amdgpu_buffer_rsrc_t rx = s_rsrc(x, g.x_bytes);
  const __amdgpu_buffer_rsrc_t rw = s_rsrc(reinterpret_cast<const unsigned char*>(wp) + (size_t)wset * g.wgs_bytes, g.w_bytes);

  // per-tap byte offset of this lane's source pixel (channel 0 + kgrp*8), SOOB outside the image / tile
  unsigned xoff[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int p = pbase + m * 32;
    const int hw = g.hout * g.wout;
    const int img = p / hw, rem = p - img * hw;
    const int oy = rem / g.wout, ox = rem - oy * g.wout;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int ky = t / (NT == 1 ? 1 : 3), kx = t - ky * (NT == 1 ? 1 : 3);
      const int iy = oy + ky - g.pad_t, ix = ox + kx - g.pad_l;
      const bool ok = p < npix && iy >= 0 && iy < g.hin && ix >= 0 && ix < g.win;
      xoff[m][t] = ok ? (unsigned)((((img * g.hin + iy) * g.win + ix) * g.cin + kgrp * 8) * 2) : SOOB;
    }
  }
  // fragment-ordered pack (conv_mfma.hip pack_coords): [row block][K chunk][tap][32 rows][16 channels] -- a wave's fragment of
  // (tap, chunk) is one contiguous KB in lane order
  const unsigned woff = (unsigned)((n0 >> 5) * g.nchunks) * (NT * 1024u) + (unsigned)((l31 * 16 + kgrp * 8) * 2);      // + (chunk * NT + tap) * 1024

  f32x16 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[m][j] = 0.f;

  // NOTE: a cin that is not a multiple of 16 (264) makes the last chunk read 8 channels of the next pixel;
  // the weight pack is zero there.
  //
  // K loop: wave w owns the chunk groups w, w+4, ... (a group = U consecutive 16-channel chunks).  The
  // fragments of group i+1 are requested before the MFMAs of group i run (two register stages), and a
  // scheduling barrier keeps the compiler from sinking those loads back between the MFMAs -- left alone
  // it keeps 2-4 loads in flight and the kernel runs at one L2 latency per load.
  struct Stage {
    bf16x8 w[U][NT];
    bf16x8 x[U][MT][NT];
  };
  const int ngroups = (g.nchunks + U - 1) / U;
  auto load = [&](Stage& st, int grp) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ck = grp * U + u;
      const bool in = ck < g.nchunks;                 // chunks past the end read zeros
      const unsigned c2 = (unsigned)(ck * 32);        // byte offset of the chunk's first channel
#pragma unroll
      for (int t = 0; t < NT; ++t) st.w[u][t] = s_load16(rw, in ? woff + (unsigned)((ck * NT + t) * 1024) : SOOB);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) st.x[u][m][t] = s_load16(rx, in ? xoff[m][t] + c2 : SOOB);
    }
  };
  auto compute = [&](const Stage& st) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m)
          acc[m] = mfma_32x32x16<F16>(st.w[u][t], st.x[u][m][t], acc[m]);
  };
  constexpr bool DB = (1 + MT) * NT * U <= 27;      // two stages fit the register file
  if constexpr (DB) {
    Stage sa, sb;
    int grp = wid;
    if (grp < ngroups) {
      load(sa, grp);
      while (true) {
        if (grp + 4 < ngroups) load(sb, grp + 4);
        __builtin_amdgcn_sched_barrier(0);
        compute(sa);
        __builtin_amdgcn_sched_barrier(0);
        grp += 4;
        if (grp >= ngroups) break;
        if (grp + 4 < ngroups) load(sa, grp + 4);
        __builtin_amdgcn_sched_barrier(0);
        compute(sb);
        __builtin_amdgcn_sched_barrier(0);
        grp += 4;
        if (grp >= ngroups) break;
      }
    }
  } else {
    Stage sa;
    for (int grp = wid; grp < ngroups; grp += 4) {
      load(sa, grp);
      __builtin_amdgcn_sched_barrier(0);
      compute(sa);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- sum the 4 K-slices, one column block at a time
  const __amdgpu_buffer_rsrc_t rbias = s_rsrc(bias, (g.epilogue & TG_EPI_BIAS) ? (unsigned)(g.cout * 4) : 0u);
  const __amdgpu_buffer_rsrc_t ry = s_rsrc(y, g.y_bytes);
  const __amdgpu_buffer_rsrc_t rmask = s_rsrc(g.mask ? (const void*)g.mask : (const void*)y, g.mask ? g.y_bytes : 0u);
  // wave w finishes register quads q = w (channels 8w + 4*kgrp .. +3 of the 32-block) for every pixel
  // -> after the half-wave swap each lane stores 8 consecutive channels (16 bytes)
  const int q = wid;
  const f32x4 bq = __builtin_bit_cast(
      f32x4, __builtin_amdgcn_raw_buffer_load_b128(rbias, (unsigned)((n0 + q * 8 + kgrp * 4) * 4), 0, 0));
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    if (m) __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wid][r][lane] = acc[m][r];
    __syncthreads();
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = q * 4 + j;
      v[j] = red[0][r][lane] + red[1][r][lane] + red[2][r][lane] + red[3][r][lane] + bq[j];
      if (g.epilogue & TG_EPI_LRELU) v[j] = lrelu_f(v[j], g.alpha);
    }
    if (g.mask) {      // uniform: a positive bf16 / f16 is a positive int16 pattern
      typedef __attribute__((ext_vector_type(2))) unsigned su32x2;
      const int pm = pbase + m * 32, chq = n0 + q * 8 + kgrp * 4;
      const su32x2 z = __builtin_bit_cast(su32x2, __builtin_amdgcn_raw_buffer_load_b64(
          rmask, (pm < npix && chq + 4 <= g.cout) ? (unsigned)((pm * g.cout + chq) * 2) : SOOB, 0, 0));
      v[0] *= (short)(z[0] & 0xffffu) > 0 ? 1.f : g.alpha;
      v[1] *= (short)(z[0] >> 16) > 0 ? 1.f : g.alpha;
      v[2] *= (short)(z[1] & 0xffffu) > 0 ? 1.f : g.alpha;
      v[3] *= (short)(z[1] >> 16) > 0 ? 1.f : g.alpha;
    }
    const unsigned p0 = pack16x2<F16>(v[0], v[1]), p1 = pack16x2<F16>(v[2], v[3]);
    if constexpr (STATS) {
      // a 4x4 image is 16 consecutive pixel lanes of a column block: a butterfly over them (fixed order) leaves the
      // image's sums of this lane's 4 channels in every lane; the first lane of each image writes them
      float r4[4] = {unpack16_lo<F16>(p0), unpack16_hi<F16>(p0), unpack16_lo<F16>(p1), unpack16_hi<F16>(p1)};
      float sq[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) sq[j] = r4[j] * r4[j];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
          r4[j] += __shfl_xor(r4[j], o, 64);
          sq[j] += __shfl_xor(sq[j], o, 64);
        }
      }
      const int pp = pbase + m * 32;
      if ((l31 & 15) == 0 && pp < npix) {
        float* out = g.stats + (size_t)(pp >> 4) * 2 * g.cout + n0 + q * 8 + kgrp * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          out[j] = r4[j];
          out[g.cout + j] = sq[j];
        }
      }
    }
    // low lanes hold channels 8q..8q+3, high lanes 8q+4..8q+7 of the same pixel: give the low lane all 8
    auto s0 = __builtin_amdgcn_permlane32_swap(p0, p0, false, false);   // s0[1] on a low lane = partner's p0
    auto s1 = __builtin_amdgcn_permlane32_swap(p1, p1, false, false);
    su32x4 o;
    o[0] = p0; o[1] = p1; o[2] = s0[1]; o[3] = s1[1];
    const int ch0 = n0 + q * 8;
    const int p = pbase + m * 32;
    const bool ok = kgrp == 0 && p < npix && ch0 + 8 <= g.cout;
    __builtin_amdgcn_raw_buffer_store_b128(o, ry, ok ? (unsigned)((p * g.cout + ch0) * 2) : SOOB, 0, TG_STORE_AUX);
  }
}

}  // namespace

// M = n*hout*wout pixels.  Worth it when the tile kernel does not apply and the pixel count is small.
bool tg_conv_small_supported(int n, int hout, int wout, int kh, int kw) {
  if (!((kh == 1 && kw == 1) || (kh == 3 && kw == 3))) return false;
  return (int64_t)n * hout * wout <= 4096;
}

// the statistics epilogue exists for 3x3 SAME convs over 4x4 maps (an image = 16 lanes of a column block), whole 32-channel
// output blocks, plain epilogue
bool tg_conv_small_stats_supported(int n, int hin, int win, int hout, int wout, int cout, int k, int pad_t, int pad_l) {
  return k == 3 && pad_t == 1 && pad_l == 1 && hin == 4 && win == 4 && hout == 4 && wout == 4 && cout % 32 == 0 &&
         tg_conv_small_supported(n, hout, wout, k, k);
}

int tg_conv_small_run(int n, int hin, int win, int cin, int hout, int wout, int cout, int k, int pad_t, int pad_l,
                      int epilogue, float alpha, const void* x, const void* wp, const float* bias, void* y,
                      hipStream_t s, float* stats, const void* mask, int groups, size_t wset_elems) {
  SmallGeom g;
  TG_CHECK(groups <= 1 || !stats, TG_ENOSUP, "conv_small: no statistics epilogue with weight-set groups");
  g.stats = stats;
  g.mask = (const bf16*)mask;
  TG_CHECK(!(mask && (stats || epilogue)), TG_ENOSUP, "conv_small: the mask epilogue comes with the plain epilogue only");
  g.n = n; g.hin = hin; g.win = win; g.cin = cin; g.hout = hout; g.wout = wout; g.cout = cout;
  g.cin_pad = (cin + 15) / 16 * 16;
  g.kh = g.kw = k;
  g.pad_t = pad_t; g.pad_l = pad_l;
  g.npix = n * hout * wout;
  g.nchunks = g.cin_pad / 16;
  g.epilogue = epilogue;
  g.alpha = alpha;
  const size_t xb = (size_t)n * hin * win * cin * 2, yb = (size_t)g.npix * cout * 2;
  const size_t rows_pad = (size_t)(cout + 63) / 64 * 64;
  const size_t wb = rows_pad * k * k * g.cin_pad * 2;
  TG_CHECK(xb < 0x7fffffffull && yb < 0x7fffffffull && wb < 0x7fffffffull, TG_ENOSUP, "conv_small: tensor too large");
  g.x_bytes = (unsigned)xb; g.y_bytes = (unsigned)yb; g.w_bytes = (unsigned)wb;
  // more pixel columns per workgroup once there are enough pixels to keep >= 256 workgroups: each weight
  // fragment then feeds MT MFMAs (the weights are the bulk of the L2 traffic of these layers)
  const int ny = (cout + 31) / 32;
  const int mt = (g.npix / 128) * ny >= 256 ? 4 : ((g.npix / 64) * ny >= 256 ? 2 : 1);
  dim3 grid((g.npix + 32 * mt - 1) / (32 * mt), ny);
  g.bpg = g.gpix = 0;
  g.wgs_bytes = 0;
  if (groups > 1) {      // every group its own workgroups: ceil(pixels of a group / pixels of a workgroup) each
    g.gpix = g.npix / groups;
    g.bpg = (g.gpix + 32 * mt - 1) / (32 * mt);
    g.wgs_bytes = (unsigned)(wset_elems * 2);
    grid.x = groups * g.bpg;
  }
#define TG_SMALL_LAUNCH(NT_, MT_)                             \
  tg_note_kernel(tg_elem_f16() ? "conv_small_kernel<%d,%d,f16%s>" : "conv_small_kernel<%d,%d%s>", NT_, MT_, g.bpg ? ",sets" : ""); \
  if (tg_elem_f16()) hipLaunchKernelGGL((conv_small_kernel<NT_, MT_, (NT_ == 1 ? 8 / MT_ : 1), true>), grid, dim3(256), 0, s, (const bf16*)x, (const bf16*)wp, bias, (bf16*)y, g); \
  else hipLaunchKernelGGL((conv_small_kernel<NT_, MT_, (NT_ == 1 ? 8 / MT_ : 1)>), grid, dim3(256), 0, s, (const bf16*)x, (const bf16*)wp, bias, (bf16*)y, g)
#define TG_SMALL_LAUNCH_STATS(MT_)                             \
  tg_note_kernel(tg_elem_f16() ? "conv_small_kernel<9,%d,f16,stats>" : "conv_small_kernel<9,%d,stats>", MT_); \
  if (tg_elem_f16()) hipLaunchKernelGGL((conv_small_kernel<9, MT_, 1, true, true>), grid, dim3(256), 0, s, (const bf16*)x, (const bf16*)wp, bias, (bf16*)y, g); \
  else hipLaunchKernelGGL((conv_small_kernel<9, MT_, 1, false, true>), grid, dim3(256), 0, s, (const bf16*)x, (const bf16*)wp, bias, (bf16*)y, g)
  if (stats) {
    TG_CHECK(tg_conv_small_stats_supported(n, hin, win, hout, wout, cout, k, pad_t, pad_l) && epilogue == 0, TG_ENOSUP,
             "conv_small: the statistics epilogue takes 3x3 SAME convs over 4x4 maps with the plain epilogue");
    if (mt == 4) { TG_SMALL_LAUNCH_STATS(4); } else if (mt == 2) { TG_SMALL_LAUNCH_STATS(2); } else { TG_SMALL_LAUNCH_STATS(1); }
  } else if (k == 1) {
    if (mt == 4) { TG_SMALL_LAUNCH(1, 4); } else if (mt == 2) { TG_SMALL_LAUNCH(1, 2); } else { TG_SMALL_LAUNCH(1, 1); }
  } else {
    if (mt == 4) { TG_SMALL_LAUNCH(9, 4); } else if (mt == 2) { TG_SMALL_LAUNCH(9, 2); } else { TG_SMALL_LAUNCH(9, 1); }
  }
#undef TG_SMALL_LAUNCH
#undef TG_SMALL_LAUNCH_STATS
  TG_LAUNCH_CHECK("conv_small");
  return TG_OK;
}
