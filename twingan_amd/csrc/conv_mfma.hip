// LDS-tiled implicit-GEMM convolution on the gfx950 matrix cores (v_mfma_f32_32x32x16_bf16),
// bf16 activations / fp32 accumulate, NHWC.
//
//   forward / backward-data : conv_fwd_mfma   (backward-data = forward over gy with the rotated,
//                                              transposed weight pack and pad' = k-1-pad)
//   backward-weight         : conv_wgrad_mfma (split-K over pixel tiles -> fp32 slabs -> reduce)
//
// Tiling (both kernels): a workgroup of 4 waves owns BM = 128 output pixels arranged as
// TI images x TH rows x TW cols (powers of two, TW*TH*TI = 128) and stages the input *halo* tile
// [TI][TH+kh-1][TW+kw-1][KC channels] in LDS once per channel chunk; every tap of the filter then
// reads its shifted window from LDS, so HBM/L2 sees each input element ~once (+halo overlap)
// instead of kh*kw times.  LDS rows are padded to an odd number of 16-byte slots so that the
// 16-byte fragment reads of a wave are bank-conflict free (MI355X_MICROARCH.md, LDS table).
//
// MFMA operand roles are swapped (A = weights, B = pixels) so the accumulator of a lane holds
// 4 consecutive output channels of one pixel per register quad -> 8-byte bf16x4 NHWC stores.
//
// Reference call sites replaced: tf.contrib.layers.conv2d at nets/pggan_utils.py:316-320 and
// its TF gradients (Conv2DBackpropInput / Conv2DBackpropFilter).
#include "tg_common.h"

namespace {

struct Geom {
  int n, hin, win, cin;        // physical input
  int hout, wout, cout;        // physical output
  int cin_pad;                 // channels in the weight pack (multiple of 16)
  int pad_t, pad_l;
  int tw_log2, th_log2, ti_log2;
  int tiles_x, tiles_y, tiles_img;   // tiles per row / per column / image groups
  int epilogue;
  float alpha;
};

extern __shared__ __attribute__((aligned(16))) unsigned char tg_smem[];

constexpr int MAXA = 6;   // max 16-byte A-tile vectors per thread per chunk

__device__ __forceinline__ bf16x8 zero8() {
  bf16x8 z;
#pragma unroll
  for (int i = 0; i < 8; ++i) z[i] = (bf16)0.f;
  return z;
}

// ------------------------------------------------------------------------------------------------
// forward (and backward-data) kernel
// ------------------------------------------------------------------------------------------------
template <int KH, int KW, int KC, int BN, bool F16 = false>
__global__ __launch_bounds__(256) void conv_fwd_mfma(const bf16* __restrict__ x, const bf16* __restrict__ wp,
                                                     const float* __restrict__ bias, bf16* __restrict__ y,
                                                     const Geom g) {
  constexpr int NT = KH * KW;
  constexpr int VPP = KC / 8;                 // 16-byte vectors per pixel per chunk
  constexpr int PS_A = KC * 2 + 16;           // LDS bytes per halo pixel (odd # of 16 B slots)
  constexpr int RS_B = NT * KC * 2 + 16;      // LDS bytes per weight row
  constexpr int NTILE = BN / 32;
  constexpr int BVEC = BN * NT * VPP;         // B-tile vectors per chunk
  constexpr int BSLOTS = (BVEC + 255) / 256;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int TW = 1 << g.tw_log2, TH = 1 << g.th_log2, TI = 1 << g.ti_log2;
  const int HWX = TW + KW - 1, HH = TH + KH - 1;
  const int halo_px = TI * HH * HWX;
  unsigned char* sA = tg_smem;
  unsigned char* sB = tg_smem + ((halo_px * PS_A + 15) & ~15);

  // tile origin
  int t = blockIdx.x;
  const int tx = t % g.tiles_x;
  t /= g.tiles_x;
  const int ty = t % g.tiles_y;
  const int tim = t / g.tiles_y;
  const int ox0 = tx << g.tw_log2, oy0 = ty << g.th_log2, img0 = tim << g.ti_log2;
  const int n0 = blockIdx.y * BN;

  // ---- fixed A-tile slots of this thread -------------------------------------------------------
  int a_goff[MAXA];    // element offset of channel 0 of the slot's pixel (+part*8), or -1
  int a_loff[MAXA];    // LDS byte offset, or -1 when the slot does not exist
  int a_part[MAXA];
  const int avec = halo_px * VPP;
#pragma unroll
  for (int s = 0; s < MAXA; ++s) {
    const int v = tid + s * 256;
    a_goff[s] = -1;
    a_loff[s] = -1;
    a_part[s] = 0;
    if (v < avec) {
      const int px = v / VPP, part = v - px * VPP;
      const int hx = px % HWX;
      const int r = px / HWX;
      const int hy = r % HH;
      const int im = r / HH;
      const int iy = oy0 + hy - g.pad_t, ix = ox0 + hx - g.pad_l, in_ = img0 + im;
      a_loff[s] = px * PS_A + part * 16;
      a_part[s] = part * 8;
      if (iy >= 0 && iy < g.hin && ix >= 0 && ix < g.win && in_ < g.n)
        a_goff[s] = ((in_ * g.hin + iy) * g.win + ix) * g.cin + part * 8;
    }
  }

  // ---- this lane's pixel (B operand / output column) -------------------------------------------
  const int m = wid * 32 + (lane & 31);
  const int mx = m & (TW - 1), my = (m >> g.tw_log2) & (TH - 1), mi = m >> (g.tw_log2 + g.th_log2);
  const int kgrp = lane >> 5;
  const int a_base = ((mi * HH + my) * HWX + mx) * PS_A + kgrp * 16;
  const int b_base = (lane & 31) * RS_B + kgrp * 16;

  f32x16 acc[NTILE];
#pragma unroll
  for (int i = 0; i < NTILE; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;

  const int wrow = NT * g.cin_pad;    // elements per packed weight row
  for (int c0 = 0; c0 < g.cin_pad; c0 += KC) {
    if (c0) __syncthreads();
    // A halo tile
#pragma unroll
    for (int s = 0; s < MAXA; ++s) {
      if (a_loff[s] >= 0) {
        bf16x8 v = zero8();
        if (a_goff[s] >= 0 && c0 + a_part[s] + 8 <= g.cin) v = *reinterpret_cast<const bf16x8*>(x + a_goff[s] + c0);
        *reinterpret_cast<bf16x8*>(sA + a_loff[s]) = v;
      }
    }
    // B (weight) tile: rows n0..n0+BN, all taps, channels c0..c0+KC
#pragma unroll
    for (int s = 0; s < BSLOTS; ++s) {
      const int v = tid + s * 256;
      if (v < BVEC) {
        const int row = v / (NT * VPP);
        const int rem = v - row * (NT * VPP);
        const int tap = rem / VPP, part = rem - tap * VPP;
        const bf16x8 w8 = *reinterpret_cast<const bf16x8*>(wp + (size_t)(n0 + row) * wrow + tap * g.cin_pad + c0 + part * 8);
        *reinterpret_cast<bf16x8*>(sB + row * RS_B + (tap * KC + part * 8) * 2) = w8;
      }
    }
    __syncthreads();
#pragma unroll
    for (int ky = 0; ky < KH; ++ky) {
#pragma unroll
      for (int kx = 0; kx < KW; ++kx) {
        const int a_tap = a_base + (ky * HWX + kx) * PS_A;
        const int tap = ky * KW + kx;
#pragma unroll
        for (int kk = 0; kk < KC / 16; ++kk) {
          const bf16x8 xf = *reinterpret_cast<const bf16x8*>(sA + a_tap + kk * 32);
#pragma unroll
          for (int nt = 0; nt < NTILE; ++nt) {
            const bf16x8 wf = *reinterpret_cast<const bf16x8*>(sB + b_base + nt * 32 * RS_B + (tap * KC + kk * 16) * 2);
            acc[nt] = mfma_32x32x16<F16>(wf, xf, acc[nt]);
          }
        }
      }
    }
  }

  // ---- epilogue: lane holds pixel m, channels n0 + nt*32 + 8*q + 4*kgrp + {0..3} ----------------
  const int oy = oy0 + my, ox = ox0 + mx, on = img0 + mi;
  if (oy < g.hout && ox < g.wout && on < g.n) {
    bf16* yp = y + ((size_t)(on * g.hout + oy) * g.wout + ox) * g.cout;
#pragma unroll
    for (int nt = 0; nt < NTILE; ++nt) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ch = n0 + nt * 32 + q * 8 + kgrp * 4;
        if (ch < g.cout) {
          float o[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float v = acc[nt][q * 4 + j];
            if (g.epilogue & TG_EPI_BIAS) v += bias[ch + j];
            if (g.epilogue & TG_EPI_LRELU) v = lrelu_f(v, g.alpha);
            o[j] = v;
          }
          typedef __attribute__((ext_vector_type(2))) unsigned u2;
          u2 pk;
          pk[0] = pack16x2<F16>(o[0], o[1]);
          pk[1] = pack16x2<F16>(o[2], o[3]);
          *reinterpret_cast<u2*>(yp + ch) = pk;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward-weight kernel.  Block = (k-slice of pixel tiles, 32-wide ci tile, 32-wide co tile).
// The 4 waves split the 128 pixels of each tile (K dimension) and keep one 32x32 fp32
// accumulator per tap; at the end they are summed through LDS and written to this k-slice's slab
// [tap][cin][cout]; conv_wgrad_reduce sums the slabs into the HWIO gradient.
// D[m = ci][n = co] += sum_pix X[pix + tap][ci] * GY[pix][co]
// ------------------------------------------------------------------------------------------------
template <int KH, int KW, bool F16 = false>
__global__ __launch_bounds__(256) void conv_wgrad_mfma(const bf16* __restrict__ x, const bf16* __restrict__ gy,
                                                       float* __restrict__ slab, const Geom g, int n_co_tiles,
                                                       int tiles_per_block, int total_tiles) {
  constexpr int NT = KH * KW;
  constexpr int PS_X = 32 * 2 + 16;   // 32 input channels per halo pixel
  constexpr int PS_G = 32 * 2 + 16;   // 32 output channels per pixel
  constexpr int MAXX = 5;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int TW = 1 << g.tw_log2, TH = 1 << g.th_log2, TI = 1 << g.ti_log2;
  const int HWX = TW + KW - 1, HH = TH + KH - 1;
  const int halo_px = TI * HH * HWX;
  unsigned char* sX = tg_smem;
  unsigned char* sG = tg_smem + ((halo_px * PS_X + 15) & ~15);

  const int ci_tile = blockIdx.y / n_co_tiles, co_tile = blockIdx.y - ci_tile * n_co_tiles;
  const int ci0 = ci_tile * 32, co0 = co_tile * 32;

  // fixed halo decomposition of this thread's X slots (tile independent)
  int x_loff[MAXX], x_hy[MAXX], x_hx[MAXX], x_im[MAXX], x_ch[MAXX];
  const int xvec = halo_px * 4;
#pragma unroll
  for (int s = 0; s < MAXX; ++s) {
    const int v = tid + s * 256;
    x_loff[s] = -1;
    x_hy[s] = x_hx[s] = x_im[s] = x_ch[s] = 0;
    if (v < xvec) {
      const int px = v >> 2, part = v & 3;
      x_hx[s] = px % HWX;
      const int r = px / HWX;
      x_hy[s] = r % HH;
      x_im[s] = r / HH;
      x_ch[s] = ci0 + part * 8;
      x_loff[s] = px * PS_X + part * 16;
    }
  }

  f32x16 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;

  const int kgrp = lane >> 5, idx = lane & 31;
  const int tile_begin = blockIdx.x * tiles_per_block;
  int tile_end = tile_begin + tiles_per_block;
  if (tile_end > total_tiles) tile_end = total_tiles;

  for (int tile = tile_begin; tile < tile_end; ++tile) {
    int t = tile;
    const int tx = t % g.tiles_x;
    t /= g.tiles_x;
    const int ty = t % g.tiles_y;
    const int tim = t / g.tiles_y;
    const int ox0 = tx << g.tw_log2, oy0 = ty << g.th_log2, img0 = tim << g.ti_log2;
    if (tile != tile_begin) __syncthreads();
    // X halo tile (32 channels from ci0)
#pragma unroll
    for (int s = 0; s < MAXX; ++s) {
      if (x_loff[s] >= 0) {
        const int iy = oy0 + x_hy[s] - g.pad_t, ix = ox0 + x_hx[s] - g.pad_l, in_ = img0 + x_im[s];
        bf16x8 v = zero8();
        if (iy >= 0 && iy < g.hin && ix >= 0 && ix < g.win && in_ < g.n && x_ch[s] + 8 <= g.cin)
          v = *reinterpret_cast<const bf16x8*>(x + ((size_t)(in_ * g.hin + iy) * g.win + ix) * g.cin + x_ch[s]);
        *reinterpret_cast<bf16x8*>(sX + x_loff[s]) = v;
      }
    }
    // GY tile: 128 pixels x 32 channels from co0 (2 vectors per thread)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int v = tid + s * 256;
      const int pm = v >> 2, part = v & 3;
      const int px_ = pm & (TW - 1), py_ = (pm >> g.tw_log2) & (TH - 1), pi_ = pm >> (g.tw_log2 + g.th_log2);
      const int oy = oy0 + py_, ox = ox0 + px_, on = img0 + pi_;
      bf16x8 w8 = zero8();
      if (oy < g.hout && ox < g.wout && on < g.n && co0 + part * 8 + 8 <= g.cout)
        w8 = *reinterpret_cast<const bf16x8*>(gy + ((size_t)(on * g.hout + oy) * g.wout + ox) * g.cout + co0 + part * 8);
      *reinterpret_cast<bf16x8*>(sG + pm * PS_G + part * 16) = w8;
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      // this lane's 8 reduction pixels of the k-step
      const int m0 = wid * 32 + ks * 16 + kgrp * 8;
      bf16x8 gf;
      int xoff[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int mj = m0 + j;
        gf[j] = *reinterpret_cast<const bf16*>(sG + mj * PS_G + idx * 2);
        const int jx = mj & (TW - 1), jy = (mj >> g.tw_log2) & (TH - 1), ji = mj >> (g.tw_log2 + g.th_log2);
        xoff[j] = ((ji * HH + jy) * HWX + jx) * PS_X + idx * 2;
      }
#pragma unroll
      for (int ky = 0; ky < KH; ++ky) {
#pragma unroll
        for (int kx = 0; kx < KW; ++kx) {
          const int tapoff = (ky * HWX + kx) * PS_X;
          bf16x8 xf;
#pragma unroll
          for (int j = 0; j < 8; ++j) xf[j] = *reinterpret_cast<const bf16*>(sX + xoff[j] + tapoff);
          acc[ky * KW + kx] = mfma_32x32x16<F16>(xf, gf, acc[ky * KW + kx]);
        }
      }
    }
  }

  // ---- cross-wave reduction through LDS, one tap at a time; write the slab ----------------------
  float* red = reinterpret_cast<float*>(tg_smem);    // [4 waves][16 regs][64 lanes]
  float* out = slab + (size_t)blockIdx.x * NT * g.cin * g.cout;
#pragma unroll
  for (int tap = 0; tap < NT; ++tap) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wid * 16 + r) * 64 + lane] = acc[tap][r];
    __syncthreads();
    // thread -> (lane' = tid & 63, regs (tid >> 6) * 4 .. +3)
    const int l2 = tid & 63, rq = tid >> 6;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = rq * 4 + j;
      const float sum = red[(0 * 16 + r) * 64 + l2] + red[(1 * 16 + r) * 64 + l2] + red[(2 * 16 + r) * 64 + l2] +
                        red[(3 * 16 + r) * 64 + l2];
      const int ci = ci0 + (r & 3) + 8 * (r >> 2) + 4 * (l2 >> 5);
      const int co = co0 + (l2 & 31);
      if (ci < g.cin && co < g.cout) out[((size_t)tap * g.cin + ci) * g.cout + co] = sum;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------------
// mode 0: out[co][tap][ci] = w[tap][ci][co];  mode 1: out[ci][tap'][co] = w[NT-1-tap'][ci][co]
// 16-bit store in the pack's element format
__device__ __forceinline__ void store16(bf16* p, float v, int half_fmt) {
  if (half_fmt) *reinterpret_cast<f16*>(p) = (f16)v;
  else *p = (bf16)v;
}

// Fragment-ordered packs (`frag`, the layers conv_img takes: 3x3 over the 8x8 maps): the same elements in the order the
// MFMA weight fragments are fetched -- out[row / 32][k / 16][tap][row % 32][k % 16] -- so that the 32 rows x 16 channels a
// wave loads for one (tap, K chunk) are ONE contiguous KB (64 lanes x 16 bytes in lane order) and a wave's nine taps of a
// chunk nine consecutive KB; in the [row][tap][k] order that load touched 32 separate 32-byte segments 4.6 KB apart
// (profiles/r05_g: 3.5 of conv_img's 10.6 us).  (row, tap, k) of pack element i:
__device__ __forceinline__ void pack_coords(int64_t i, int nt, int inner_pad, int frag, int* row, int* tap, int* k) {
  if (frag) {
    const int within = (int)(i & 511);
    int64_t r = i >> 9;
    *tap = (int)(r % nt);
    r /= nt;
    const int nch = inner_pad >> 4;
    *k = (int)(r % nch) * 16 + (within & 15);
    *row = (int)(r / nch) * 32 + (within >> 4);
  } else {
    *k = (int)(i % inner_pad);
    const int64_t r = i / inner_pad;
    *tap = (int)(r % nt);
    *row = (int)(r / nt);
  }
}

__global__ void pack_weights(const float* __restrict__ w, bf16* __restrict__ out, int nt, int cin, int cout, int rows,
                             int rows_pad, int inner, int inner_pad, int mode, int half_fmt, int frag) {
  const int64_t total = (int64_t)rows_pad * nt * inner_pad;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int k, tap, row;
    pack_coords(i, nt, inner_pad, frag, &row, &tap, &k);
    float v = 0.f;
    if (row < rows && k < inner) {
      if (mode == 0)
        v = w[((int64_t)tap * cin + k) * cout + row];
      else
        v = w[((int64_t)(nt - 1 - tap) * cin + row) * cout + k];
    }
    store16(out + i, v, half_fmt);
  }
}

inline int ilog2(int v) {
  int r = 0;
  while ((1 << r) < v) ++r;
  return r;
}

// choose the 128-pixel tile shape for an output of hout x wout
void pick_tile(int hout, int wout, Geom* g) {
  int tw = 1 << ilog2(wout);
  if (tw > 16) tw = 16;
  int th = 1 << ilog2(hout);
  if (th > 128 / tw) th = 128 / tw;
  const int ti = 128 / (tw * th);
  g->tw_log2 = ilog2(tw);
  g->th_log2 = ilog2(th);
  g->ti_log2 = ilog2(ti);
  g->tiles_x = (wout + tw - 1) / tw;
  g->tiles_y = (hout + th - 1) / th;
}

int fill_geom(const char* who, int n, int hin, int win, int cin, int hout, int wout, int cout, int kh, int kw, int pad_t,
              int pad_l, Geom* g) {
  TG_CHECK(cin % 8 == 0 && cout % 8 == 0, TG_EALIGN, "%s(mfma): cin (%d) and cout (%d) must be multiples of 8", who, cin,
           cout);
  TG_CHECK((kh == 1 && kw == 1) || (kh == 3 && kw == 3), TG_ENOSUP, "%s(mfma): kernel %dx%d not supported", who, kh, kw);
  g->n = n; g->hin = hin; g->win = win; g->cin = cin;
  g->hout = hout; g->wout = wout; g->cout = cout;
  g->cin_pad = (cin + 15) / 16 * 16;
  g->pad_t = pad_t; g->pad_l = pad_l;
  pick_tile(hout, wout, g);
  const int ti = 1 << g->ti_log2;
  g->tiles_img = (n + ti - 1) / ti;
  g->epilogue = 0;
  g->alpha = 1.f;
  return TG_OK;
}

template <int KH, int KW, int KC, int BN>
int launch_fwd(const Geom& g, const bf16* x, const bf16* wp, const float* bias, bf16* y, hipStream_t s) {
  const int TW = 1 << g.tw_log2, TH = 1 << g.th_log2, TI = 1 << g.ti_log2;
  const int halo_px = TI * (TH + KH - 1) * (TW + KW - 1);
  TG_CHECK(halo_px * (KC / 8) <= 256 * MAXA, TG_ENOSUP, "conv(mfma): halo tile too large (%d px)", halo_px);
  const size_t lds = ((size_t)(halo_px * (KC * 2 + 16) + 15) & ~(size_t)15) + (size_t)BN * (KH * KW * KC * 2 + 16);
  TG_CHECK(lds <= 64 * 1024, TG_ENOSUP, "conv(mfma): LDS %zu > 64 KiB", lds);
  dim3 grid(g.tiles_x * g.tiles_y * g.tiles_img, (g.cout + BN - 1) / BN);
  tg_note_kernel(tg_elem_f16() ? "conv_fwd_mfma<%d,%d,%d,%d,f16>" : "conv_fwd_mfma<%d,%d,%d,%d>", KH, KW, KC, BN);
  if (tg_elem_f16())
    hipLaunchKernelGGL((conv_fwd_mfma<KH, KW, KC, BN, true>), grid, dim3(256), lds, s, x, wp, bias, y, g);
  else
    hipLaunchKernelGGL((conv_fwd_mfma<KH, KW, KC, BN>), grid, dim3(256), lds, s, x, wp, bias, y, g);
  TG_LAUNCH_CHECK("conv_fwd_mfma");
  return TG_OK;
}

template <int KH, int KW>
int dispatch_fwd(const Geom& g, const bf16* x, const bf16* wp, const float* bias, bf16* y, hipStream_t s) {
  const bool wide = g.cout > 32;
  if constexpr (KH == 1) {
    if (g.cin_pad % 64 == 0)
    return wide ? launch_fwd<KH, KW, 64, 64>(g, x, wp, bias, y, s) : launch_fwd<KH, KW, 64, 32>(g, x, wp, bias, y, s);
  }
  if (g.cin_pad % 32 == 0) {
    return wide ? launch_fwd<KH, KW, 32, 64>(g, x, wp, bias, y, s) : launch_fwd<KH, KW, 32, 32>(g, x, wp, bias, y, s);
  }
  return wide ? launch_fwd<KH, KW, 16, 64>(g, x, wp, bias, y, s) : launch_fwd<KH, KW, 16, 32>(g, x, wp, bias, y, s);
}

}  // namespace

// Rewrites "k x k VALID on a k x k input" (1x1 output) as a 1x1 conv over k*k*cin channels: with NHWC
// activations and HWIO weights both reshapes are free (nets/pggan.py:330-331,495).
static inline bool is16(const TgConvDesc* d) { return d->dtype == TG_BF16 || d->dtype == TG_F16; }

static bool as_dense(const TgConvDesc* d, TgConvDesc* o) {
  if (d->hout == 1 && d->wout == 1 && d->hin == d->kh && d->win == d->kw && d->pad_t == 0 && d->pad_l == 0 &&
      (d->kh > 1 || d->kw > 1)) {
    *o = *d;
    o->cin = d->cin * d->kh * d->kw;
    o->hin = o->win = 1;
    o->kh = o->kw = 1;
    return true;
  }
  return false;
}

bool tg_conv_tile_supported(int h, int w, int hout, int wout, int kh, int kw, int pad_t, int pad_l);
bool tg_conv_img_supported(int n, int hin, int win, int cin, int hout, int wout, int cout, int k, int pad_t, int pad_l);
bool tg_conv_small_supported(int n, int hout, int wout, int kh, int kw);

// Is the pack of (descriptor, mode) fragment-ordered?  Exactly when the dispatch of that direction ends in conv_img or
// conv_small (tg_conv2d_fwd_mfma / tg_conv2d_bwd_data_mfma: the tile kernels first, then conv_img, then conv_small) --
// the two kernels that fetch their weight fragments straight from L2: the layout is a property of the pack that its
// kernel knows; callers treat packs as opaque.  (conv_small's test involves the batch: n * hout * wout <= 4096.)
static bool pack_frag(const TgConvDesc* d0, int mode) {
  TgConvDesc dd;
  const TgConvDesc* d = as_dense(d0, &dd) ? &dd : d0;
  if (!is16(d) || d->algo != TG_ALGO_MFMA || d->kh != d->kw || (d->kh != 3 && d->kh != 1) || d->cin % 8 || d->cout % 8) return false;
  // the conv the kernels see: forward, or the same conv over gy with the rotated pack (mode 1)
  const bool fw = mode == 0;
  const int hi = fw ? d->hin : d->hout, wi = fw ? d->win : d->wout, ci = fw ? d->cin : d->cout;
  const int ho = fw ? d->hout : d->hin, wo = fw ? d->wout : d->win, co = fw ? d->cout : d->cin;
  const int pt = fw ? d->pad_t : d->kh - 1 - d->pad_t, pl = fw ? d->pad_l : d->kw - 1 - d->pad_l;
  if (tg_conv_tile_supported(hi, wi, ho, wo, d->kh, d->kw, d->pad_t, d->pad_l)) return false;
  if (tg_conv_img_supported(d->n, hi, wi, ci, ho, wo, co, d->kh, pt, pl)) return true;
  return d->pad_t == d->pad_l && tg_conv_small_supported(d->n, ho, wo, d->kh, d->kw);      // conv_small (4x4 maps, dense layers)
}
extern "C" int tg_conv2d_pack_layout(const TgConvDesc* d, int mode) { return d && pack_frag(d, mode) ? 1 : 0; }

static void pack_dims(const TgConvDesc* d0, int mode, int* nt, int* cin, int* cout, int* rows, int* rows_pad,
                      int* inner, int* inner_pad) {
  TgConvDesc dd;
  const TgConvDesc* d = as_dense(d0, &dd) ? &dd : d0;
  *nt = d->kh * d->kw;
  *cin = d->cin;
  *cout = d->cout;
  *rows = mode == 0 ? d->cout : d->cin;
  *inner = mode == 0 ? d->cin : d->cout;
  *rows_pad = (*rows + 63) / 64 * 64;      // a BN=64 block never reads past the pack
  *inner_pad = (*inner + 15) / 16 * 16;
}

size_t tg_conv2d_pack_elems(const TgConvDesc* d, int mode) {
  int nt, cin, cout, rows, rows_pad, inner, inner_pad;
  pack_dims(d, mode, &nt, &cin, &cout, &rows, &rows_pad, &inner, &inner_pad);
  return (size_t)rows_pad * nt * inner_pad * (d->groups > 1 ? d->groups : 1);      // one pack per weight set, back to back
}

int tg_conv2d_pack_weights(const TgConvDesc* d, const float* w, int mode, void* out, void* stream) {
  TG_CHECK(mode == 0 || mode == 1, TG_EINVAL, "tg_conv2d_pack_weights: mode %d", mode);
  int nt, cin, cout, rows, rows_pad, inner, inner_pad;
  pack_dims(d, mode, &nt, &cin, &cout, &rows, &rows_pad, &inner, &inner_pad);
  const int64_t total = (int64_t)rows_pad * nt * inner_pad;
  for (int g = 0; g < (d->groups > 1 ? d->groups : 1); ++g) {      // weight set g: master w[g] -> pack g
    hipLaunchKernelGGL(pack_weights, dim3(tg_grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       w + (size_t)g * nt * cin * cout, (bf16*)out + (size_t)g * total, nt, cin, cout, rows, rows_pad, inner,
                       inner_pad, mode, d->dtype == TG_F16 ? 1 : 0, pack_frag(d, mode) ? 1 : 0);
    TG_LAUNCH_CHECK("tg_conv2d_pack_weights");
  }
  return TG_OK;
}

// ---- multi-tensor pack: one launch for every pack of an optimiser group
namespace {
struct PackJob {
  const float* w;
  bf16* out;
  int nt, cin, cout, rows, rows_pad, inner, inner_pad, mode;
  int half_fmt;                    // element format of this pack: 1 = IEEE half (TG_F16 descriptors)
  int frag;                        // fragment-ordered pack (pack_coords)
  int block_begin, block_end;      // this job's slice of the grid (PACK_EPB elements per block)
};
constexpr int PACK_EPB = 2048;

__global__ __launch_bounds__(256) void pack_weights_multi(const PackJob* __restrict__ jobs, int njobs) {
  __shared__ float tile[64][65];
  // binary search: the job whose [block_begin, block_end) holds this block (uniform -> scalar loads)
  int lo = 0, hi = njobs - 1;
  const int b = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (b >= jobs[mid].block_end) lo = mid + 1;
    else hi = mid;
  }
  const PackJob j = jobs[lo];
  if (j.mode == 0) {
    // out[co][tap][ci] = w[tap][ci][co]: a transpose per tap.  One block = one 64 (ci) x 64 (co) tile through LDS so
    // that both the fp32 reads (along co) and the bf16 writes (along ci) are coalesced.
    const int kt = (j.inner_pad + 63) / 64, rt = j.rows_pad / 64;
    int t = b - j.block_begin;
    const int k0 = (t % kt) * 64;
    t /= kt;
    const int r0 = (t % rt) * 64;
    const int tap = t / rt;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll 4
    for (int i = ty; i < 64; i += 4) {
      const int ci = k0 + i, co = r0 + tx;
      tile[i][tx] = (ci < j.cin && co < j.cout) ? j.w[((int64_t)tap * j.cin + ci) * j.cout + co] : 0.f;
    }
    __syncthreads();
    if (j.frag) {
      // the 64 (ci) x 64 (co) tile = 2 row blocks x 4 K chunks of this tap: eight contiguous 512-element runs of the pack
      // (rows_pad % 64 == 0 always; cin a multiple of 32: chunks past the kernel's last one are skipped)
      const int nch = j.inner_pad >> 4;
#pragma unroll 4
      for (int e = threadIdx.x; e < 4096; e += 256) {
        const int run = e >> 9, within = e & 511;
        const int co_l = (run >> 2) * 32 + (within >> 4), ci_l = (run & 3) * 16 + (within & 15);
        const int64_t o = ((((int64_t)(r0 >> 5) + (run >> 2)) * nch + (k0 >> 4) + (run & 3)) * j.nt + tap) * 512 + within;
        if ((k0 >> 4) + (run & 3) < nch) store16(j.out + o, tile[ci_l][co_l], j.half_fmt);      // a 32-channel kernel: two chunks
      }
      return;
    }
#pragma unroll 4
    for (int i = ty; i < 64; i += 4) {
      const int co = r0 + i, ci = k0 + tx;
      if (ci < j.inner_pad) store16(j.out + ((int64_t)co * j.nt + tap) * j.inner_pad + ci, tile[tx][i], j.half_fmt);
    }
    return;
  }
  const int64_t total = (int64_t)j.rows_pad * j.nt * j.inner_pad;
  const int64_t i0 = (int64_t)(b - j.block_begin) * PACK_EPB;
#pragma unroll 2
  for (int e = threadIdx.x; e < PACK_EPB; e += 256) {
    const int64_t i = i0 + e;
    if (i >= total) break;
    int k, tap, row;
    pack_coords(i, j.nt, j.inner_pad, j.frag, &row, &tap, &k);
    float v = 0.f;
    if (row < j.rows && k < j.inner) v = j.w[((int64_t)(j.nt - 1 - tap) * j.cin + row) * j.cout + k];      // mode 1
    store16(j.out + i, v, j.half_fmt);
  }
}
}  // namespace

size_t tg_pack_table_bytes(int njobs) { return (size_t)(njobs > 0 ? njobs : 0) * sizeof(PackJob); }

int tg_pack_table_fill(const TgConvDesc* d, const float* w, int mode, void* out, int job, void* table_host,
                       int32_t* total_blocks) {
  TG_CHECK(d && w && out && table_host && total_blocks && job >= 0, TG_EINVAL, "tg_pack_table_fill: bad arguments");
  TG_CHECK(mode == 0 || mode == 1, TG_EINVAL, "tg_pack_table_fill: mode %d", mode);
  TG_CHECK(d->groups <= 1, TG_EINVAL, "tg_pack_table_fill: one job per weight set (fill each set of a grouped descriptor as its own job)");
  PackJob j;
  j.w = w;
  j.out = (bf16*)out;
  j.mode = mode;
  j.half_fmt = d->dtype == TG_F16;
  j.frag = pack_frag(d, mode) ? 1 : 0;
  pack_dims(d, mode, &j.nt, &j.cin, &j.cout, &j.rows, &j.rows_pad, &j.inner, &j.inner_pad);
  TG_CHECK(!j.frag || (j.inner_pad % 16 == 0 && j.rows_pad % 64 == 0), TG_ENOSUP, "tg_pack_table_fill: fragment-ordered pack of a %d x %d kernel", j.rows_pad, j.inner_pad);
  const int64_t total = (int64_t)j.rows_pad * j.nt * j.inner_pad;
  j.block_begin = *total_blocks;
  if (mode == 0)      // one block per (tap, 64 rows, 64 inner) tile
    j.block_end = j.block_begin + j.nt * (j.rows_pad / 64) * ((j.inner_pad + 63) / 64);
  else
    j.block_end = j.block_begin + (int)((total + PACK_EPB - 1) / PACK_EPB);
  *total_blocks = j.block_end;
  ((PackJob*)table_host)[job] = j;
  return TG_OK;
}

int tg_conv2d_pack_weights_multi(const void* table_device, int njobs, int total_blocks, void* stream) {
  TG_CHECK(table_device && njobs > 0 && total_blocks > 0, TG_EINVAL, "tg_conv2d_pack_weights_multi: bad arguments");
  hipLaunchKernelGGL(pack_weights_multi, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream,
                     (const PackJob*)table_device, njobs);
  TG_LAUNCH_CHECK("tg_conv2d_pack_weights_multi");
  return TG_OK;
}

bool tg_conv_tile_supported(int h, int w, int hout, int wout, int kh, int kw, int pad_t, int pad_l);
int tg_conv_tile_run(int n, int h, int w, int cin, int cout, int k, int pad, int epilogue, float alpha, const void* x,
                     const void* wp, const float* bias, void* y, hipStream_t s, const void* mask = nullptr,
                     float* stats = nullptr, int stat_chunks = 0, int* chunks_query = nullptr, void* ypool = nullptr,
                     void* ymask = nullptr, const void* up_src = nullptr, const void* up_signs = nullptr, float up_alpha = 0.f,
                     void* up_store = nullptr, const void* up_z = nullptr, int groups = 1, size_t wset_elems = 0);
bool tg_conv_tile_grouped_native(int n, int h, int w, int cin, int cout);

// Grouped calls (TgConvDesc::groups > 1) the dispatch below takes as ONE launch: the kernel picks the weight set from the
// image index.  op: 0 forward-shaped (forward, masked, pool), 1 backward-data-shaped, 2 filter gradient.  Everything else
// is launched once per group by the entry point (capi.hip).
bool tg_conv_img_supported(int n, int hin, int win, int cin, int hout, int wout, int cout, int k, int pad_t, int pad_l);
bool tg_conv_small_supported(int n, int hout, int wout, int kh, int kw);
bool tg_conv2d_grouped_native_mfma(const TgConvDesc* d0, int op) {
  TgConvDesc dd;
  const TgConvDesc* d = as_dense(d0, &dd) ? &dd : d0;
  if (!is16(d) || d->algo == TG_ALGO_MFMA_V1 || d->cin % 8 || d->cout % 8 || d->groups < 2) return false;
  if (op == 2) return false;      // filter gradients: one launch per group
  // the conv the kernels see: forward (x -> y) or the same conv over gy with the rotated pack (gy -> gx)
  const bool fw = op == 0;
  const int hi = fw ? d->hin : d->hout, wi = fw ? d->win : d->wout, ci = fw ? d->cin : d->cout;
  const int ho = fw ? d->hout : d->hin, wo = fw ? d->wout : d->win, co = fw ? d->cout : d->cin;
  const int pt = fw ? d->pad_t : d->kh - 1 - d->pad_t, pl = fw ? d->pad_l : d->kw - 1 - d->pad_l;
  if (tg_conv_tile_supported(hi, wi, ho, wo, d->kh, d->kw, d->pad_t, d->pad_l)) return tg_conv_tile_grouped_native(d->n, hi, wi, ci, co);
  if (d->kh == d->kw && tg_conv_img_supported(d->n, hi, wi, ci, ho, wo, co, d->kh, pt, pl)) return true;
  return d->pad_t == d->pad_l && tg_conv_small_supported(d->n, ho, wo, d->kh, d->kw);
}
// weight sets of a descriptor and the elements of one set's pack (mode 0 forward / 1 backward-data operand)
static int ngroups(const TgConvDesc* d) { return d->groups > 1 ? d->groups : 1; }
static size_t wset_elems(const TgConvDesc* d, int mode) {
  if (d->groups <= 1) return 0;
  TgConvDesc d1 = *d;
  d1.groups = 1;
  return tg_conv2d_pack_elems(&d1, mode);
}

// Forward conv that also writes the 2x2 average pool of its output (conv_tile.hip POOL kernels): 3x3 SAME, even h / w,
// shapes the tile kernels take
bool tg_conv2d_fwd_pool_supported_mfma(const TgConvDesc* d0) {
  TgConvDesc dd;
  const TgConvDesc* d = as_dense(d0, &dd) ? &dd : d0;
  if (!is16(d) || d->algo == TG_ALGO_MFMA_V1 || d->kh != 3 || d->cin % 8 || d->cout % 8) return false;
  return tg_conv_tile_supported(d->hin, d->win, d->hout, d->wout, d->kh, d->kw, d->pad_t, d->pad_l);
}

int tg_conv2d_fwd_pool_mfma(const TgConvDesc* d0, const void* x, const void* wp, const float* bias, void* y, void* ypool,
                            hipStream_t s, void* ymask) {
  TgConvDesc dd;
  const TgConvDesc* d = as_dense(d0, &dd) ? &dd : d0;
  TG_CHECK(tg_conv2d_fwd_pool_supported_mfma(d0), TG_ENOSUP, "tg_conv2d_fwd_pool: shape not taken (query tg_conv2d_fwd_pool_supported)");
  TG_CHECK(!(d->epilogue & TG_EPI_BIAS) || bias, TG_EINVAL, "tg_conv2d_fwd_pool: bias epilogue without bias pointer");
  return tg_conv_tile_run(d->n, d->hin, d->win, d->cin, d->cout, d->kh, d->pad_t, d->epilogue, d->lrelu_alpha, x, wp, bias, y,
                          s, nullptr, nullptr, 0, nullptr, ypool, ymask, nullptr, nullptr, 0.f, nullptr, nullptr, ngroups(d0),
                          wset_elems(d0, 0));
}

// Forward conv that also writes the per-workgroup statistics partials of its output (conv_tile.hip STATS kernels).
// chunks per image of the dispatch this descriptor selects, 0 when that dispatch has no statistics epilogue.
bool tg_conv_small_supported(int n, int hout, int wout, int kh, int kw);
bool tg_conv_small_stats_supported(int n, int hin, int win, int hout, int wout, int cout, int k, int pad_t, int pad_l);
int tg_conv_small_run(int n, int hin, int win, int cin, int hout, int wout, int cout, int k, int pad_t, int pad_l,
                      int epilogue, float alpha, const void* x, const void* wp, const float* bias, void* y, hipStream_t s,
                      float* stats = nullptr, const void* mask = nullptr, int groups = 1, size_t wset_elems = 0);
bool tg_conv_img_supported(int n, int hin, int win, int cin, int hout, int wout, int cout, int k, int pad_t, int pad_l);
bool tg_conv_img_stats_supported(int n, int hin, int win, int cin, int hout, int wout, int cout, int k, int pad_t, int pad_l);
int tg_conv_img_run(int n, int hw, int cin, int cout, int epilogue, float alpha, const void* x, const void* wp,
                    const float* bias, void* y, hipStream_t s, float* stats = nullptr, const void* mask = nullptr, int groups = 1,
                    size_t wset_elems = 0);

int tg_conv2d_fwd_stats_chunks_mfma(const TgConvDesc* d0) {
  TgConvDesc dd;
  const TgConvDesc* d = as_dense(d0, &dd) ? &dd : d0;
  if (!is16(d) || d->algo == TG_ALGO_MFMA_V1 || d->kh != 3 || d->cin % 8 || d->cout % 8 || d->epilogue) return 0;
  if (!tg_conv_tile_supported(d->hin, d->win, d->hout, d->wout, d->kh, d->kw, d->pad_t, d->pad_l)) {
    // 8x8 maps: conv_img holds a whole image per workgroup; 4x4 maps: an image is 16 lanes of conv_small's column block
    // -- ONE chunk per image (the order of the tests mirrors tg_conv2d_fwd_mfma's dispatch)
    if (d->kh == d->kw && tg_conv_img_supported(d->n, d->hin, d->win, d->cin, d->hout, d->wout, d->cout, d->kh, d->pad_t, d->pad_l))
      return tg_conv_img_stats_supported(d->n, d->hin, d->win, d->cin, d->hout, d->wout, d->cout, d->kh, d->pad_t, d->pad_l) ? 1 : 0;
    if (d->pad_t == d->pad_l && tg_conv_small_supported(d->n, d->hout, d->wout, d->kh, d->kw))
      return tg_conv_small_stats_supported(d->n, d->hin, d->win, d->hout, d->wout, d->cout, d->kh, d->pad_t, d->pad_l) ? 1 : 0;
    return 0;
  }
  int chunks = 0;
  if (tg_conv_tile_run(d->n, d->hin, d->win, d->cin, d->cout, d->kh, d->pad_t, 0, 0.f, nullptr, nullptr, nullptr, nullptr,
                       nullptr, nullptr, nullptr, 0, &chunks) != TG_OK)
    return 0;
  return chunks;
}

int tg_conv2d_fwd_stats_mfma(const TgConvDesc* d0, const void* x, const void* wp, void* y, float* partials, int chunks,
                             hipStream_t s) {
  TgConvDesc dd;
  const TgConvDesc* d = as_dense(d0, &dd) ? &dd : d0;
  TG_CHECK(chunks > 0 && chunks == tg_conv2d_fwd_stats_chunks_mfma(d0), TG_EINVAL,
           "tg_conv2d_fwd_stats: chunks %d does not match tg_conv2d_fwd_stats_chunks()", chunks);
  if (!tg_conv_tile_supported(d->hin, d->win, d->hout, d->wout, d->kh, d->kw, d->pad_t, d->pad_l)) {
    if (tg_conv_img_supported(d->n, d->hin, d->win, d->cin, d->hout, d->wout, d->cout, d->kh, d->pad_t, d->pad_l))
      return tg_conv_img_run(d->n, d->hin, d->cin, d->cout, 0, d->lrelu_alpha, x, wp, nullptr, y, s, partials);
    return tg_conv_small_run(d->n, d->hin, d->win, d->cin, d->hout, d->wout, d->cout, d->kh, d->pad_t, d->pad_l, 0,
                             d->lrelu_alpha, x, wp, nullptr, y, s, partials);
  }
  return tg_conv_tile_run(d->n, d->hin, d->win, d->cin, d->cout, d->kh, d->pad_t, 0, d->lrelu_alpha, x, wp, nullptr, y, s,
                          nullptr, partials, chunks, nullptr);
}



int tg_conv2d_fwd_mfma(const TgConvDesc* d0, const void* x, const void* wp, const float* bias, void* y, hipStream_t s) {
  TgConvDesc dd;
  const TgConvDesc* d = as_dense(d0, &dd) ? &dd : d0;
  TG_CHECK(is16(d), TG_ENOSUP, "tg_conv2d_fwd(mfma): 16-bit activations only");
  TG_CHECK(!(d->epilogue & TG_EPI_BIAS) || bias, TG_EINVAL, "tg_conv2d_fwd: bias epilogue without bias pointer");
  if (d->algo != TG_ALGO_MFMA_V1 && d->cin % 8 == 0 && d->cout % 8 == 0 &&
      tg_conv_tile_supported(d->hin, d->win, d->hout, d->wout, d->kh, d->kw, d->pad_t, d->pad_l))
    return tg_conv_tile_run(d->n, d->hin, d->win, d->cin, d->cout, d->kh, d->pad_t, d->epilogue, d->lrelu_alpha, x, wp,
                            bias, y, s, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr, nullptr,
                            ngroups(d0), wset_elems(d0, 0));
  if (d->algo != TG_ALGO_MFMA_V1 && d->kh == d->kw &&
      tg_conv_img_supported(d->n, d->hin, d->win, d->cin, d->hout, d->wout, d->cout, d->kh, d->pad_t, d->pad_l))
    return tg_conv_img_run(d->n, d->hin, d->cin, d->cout, d->epilogue, d->lrelu_alpha, x, wp, bias, y, s, nullptr, nullptr,
                           ngroups(d0), wset_elems(d0, 0));
  if (d->algo != TG_ALGO_MFMA_V1 && d->cin % 8 == 0 && d->cout % 8 == 0 && d->pad_t == d->pad_l &&
      tg_conv_small_supported(d->n, d->hout, d->wout, d->kh, d->kw))
    return tg_conv_small_run(d->n, d->hin, d->win, d->cin, d->hout, d->wout, d->cout, d->kh, d->pad_t, d->pad_l,
                             d->epilogue, d->lrelu_alpha, x, wp, bias, y, s, nullptr, nullptr, ngroups(d0), wset_elems(d0, 0));
  TG_CHECK(d0->groups <= 1, TG_ENOSUP, "tg_conv2d_fwd(mfma): weight-set groups on a first-generation kernel");
  Geom g;
  int rc = fill_geom("tg_conv2d_fwd", d->n, d->hin, d->win, d->cin, d->hout, d->wout, d->cout, d->kh, d->kw, d->pad_t,
                     d->pad_l, &g);
  if (rc) return rc;
  g.epilogue = d->epilogue;
  g.alpha = d->lrelu_alpha;
  TG_CHECK(!(d->epilogue & TG_EPI_BIAS) || bias, TG_EINVAL, "tg_conv2d_fwd: bias epilogue without bias pointer");
  if (d->kh == 1) return dispatch_fwd<1, 1>(g, (const bf16*)x, (const bf16*)wp, bias, (bf16*)y, s);
  return dispatch_fwd<3, 3>(g, (const bf16*)x, (const bf16*)wp, bias, (bf16*)y, s);
}

// Forward conv whose output is multiplied by the LeakyReLU derivative of `mask_src` (same shape as the output): the tile
// kernels' mask epilogue (built for the masked backward-data) on the forward pack.  Shapes the tile kernels take, no
// bias / activation epilogue.
bool tg_conv2d_fwd_mask_fusable_mfma(const TgConvDesc* d0) {
  TgConvDesc dd;
  const TgConvDesc* d = as_dense(d0, &dd) ? &dd : d0;
  if (!is16(d) || d->algo == TG_ALGO_MFMA_V1 || d->cin % 8 != 0 || d->cout % 8 != 0 || d->epilogue != 0) return false;
  if (tg_conv_tile_supported(d->hin, d->win, d->hout, d->wout, d->kh, d->kw, d->pad_t, d->pad_l)) return true;
  // 8x8 / 4x4 maps and the dense rewrite: conv_img / conv_small carry the same mask epilogue (tests as tg_conv2d_fwd_mfma's)
  if (d->kh == d->kw && tg_conv_img_supported(d->n, d->hin, d->win, d->cin, d->hout, d->wout, d->cout, d->kh, d->pad_t, d->pad_l))
    return true;
  return d->pad_t == d->pad_l && tg_conv_small_supported(d->n, d->hout, d->wout, d->kh, d->kw);
}

int tg_conv2d_fwd_masked_mfma(const TgConvDesc* d0, const void* x, const void* wp, const void* mask_src, void* y, hipStream_t s) {
  TgConvDesc dd;
  const TgConvDesc* d = as_dense(d0, &dd) ? &dd : d0;
  TG_CHECK(tg_conv2d_fwd_mask_fusable_mfma(d0), TG_ENOSUP, "tg_conv2d_fwd_masked(mfma): mask not fusable here");
  if (tg_conv_tile_supported(d->hin, d->win, d->hout, d->wout, d->kh, d->kw, d->pad_t, d->pad_l))
    return tg_conv_tile_run(d->n, d->hin, d->win, d->cin, d->cout, d->kh, d->pad_t, 0, d->lrelu_alpha, x, wp, nullptr, y, s,
                            mask_src, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr, nullptr, ngroups(d0),
                            wset_elems(d0, 0));
  if (d->kh == d->kw && tg_conv_img_supported(d->n, d->hin, d->win, d->cin, d->hout, d->wout, d->cout, d->kh, d->pad_t, d->pad_l))
    return tg_conv_img_run(d->n, d->hin, d->cin, d->cout, 0, d->lrelu_alpha, x, wp, nullptr, y, s, nullptr, mask_src, ngroups(d0),
                           wset_elems(d0, 0));
  return tg_conv_small_run(d->n, d->hin, d->win, d->cin, d->hout, d->wout, d->cout, d->kh, d->pad_t, d->pad_l, 0,
                           d->lrelu_alpha, x, wp, nullptr, y, s, nullptr, mask_src, ngroups(d0), wset_elems(d0, 0));
}

// Can the LeakyReLU backward of the producer of x be folded into this backward-data's epilogue?
bool tg_conv2d_bwd_data_mask_fusable_mfma(const TgConvDesc* d0) {
  TgConvDesc dd;
  const TgConvDesc* d = as_dense(d0, &dd) ? &dd : d0;
  if (!is16(d) || d->algo == TG_ALGO_MFMA_V1 || d->cin % 8 != 0 || d->cout % 8 != 0) return false;
  if (tg_conv_tile_supported(d->hout, d->wout, d->hin, d->win, d->kh, d->kw, d->pad_t, d->pad_l)) return true;
  // the 8x8 / 4x4 maps and the dense k x k VALID layers (conv_img / conv_small: the same tests as tg_conv2d_bwd_data_mfma)
  if (d->kh == d->kw && tg_conv_img_supported(d->n, d->hout, d->wout, d->cout, d->hin, d->win, d->cin, d->kh, d->kh - 1 - d->pad_t,
                                              d->kw - 1 - d->pad_l))
    return true;
  return d->pad_t == d->pad_l && tg_conv_small_supported(d->n, d->hin, d->win, d->kh, d->kw);
}

int tg_conv2d_bwd_data_mfma(const TgConvDesc* d0, const void* gy, const void* wp, void* gx, hipStream_t s,
                            const void* mask) {
  TgConvDesc dd;
  const TgConvDesc* d = as_dense(d0, &dd) ? &dd : d0;
  TG_CHECK(is16(d), TG_ENOSUP, "tg_conv2d_bwd_data(mfma): 16-bit activations only");
  TG_CHECK(!mask || tg_conv2d_bwd_data_mask_fusable_mfma(d0), TG_ENOSUP, "tg_conv2d_bwd_data(mfma): mask not fusable here");
  if (d->algo != TG_ALGO_MFMA_V1 && d->cin % 8 == 0 && d->cout % 8 == 0 &&
      tg_conv_tile_supported(d->hout, d->wout, d->hin, d->win, d->kh, d->kw, d->pad_t, d->pad_l))
    return tg_conv_tile_run(d->n, d->hout, d->wout, d->cout, d->cin, d->kh, d->kh - 1 - d->pad_t, 0,
                            mask ? d->lrelu_alpha : 1.f, gy, wp, nullptr, gx, s, mask, nullptr, 0, nullptr, nullptr, nullptr, nullptr,
                            nullptr, 0.f, nullptr, nullptr, ngroups(d0), wset_elems(d0, 1));
  // backward-data = the same conv over gy with the rotated pack: in = (hout, wout, cout), out = (hin, win, cin), pad' = k-1-pad
  if (d->algo != TG_ALGO_MFMA_V1 && d->kh == d->kw &&
      tg_conv_img_supported(d->n, d->hout, d->wout, d->cout, d->hin, d->win, d->cin, d->kh, d->kh - 1 - d->pad_t,
                            d->kw - 1 - d->pad_l))
    return tg_conv_img_run(d->n, d->hout, d->cout, d->cin, 0, mask ? d->lrelu_alpha : 1.f, gy, wp, nullptr, gx, s, nullptr, mask,
                           ngroups(d0), wset_elems(d0, 1));
  if (d->algo != TG_ALGO_MFMA_V1 && d->cin % 8 == 0 && d->cout % 8 == 0 && d->pad_t == d->pad_l &&
      tg_conv_small_supported(d->n, d->hin, d->win, d->kh, d->kw))
    return tg_conv_small_run(d->n, d->hout, d->wout, d->cout, d->hin, d->win, d->cin, d->kh, d->kh - 1 - d->pad_t,
                             d->kw - 1 - d->pad_l, 0, mask ? d->lrelu_alpha : 1.f, gy, wp, nullptr, gx, s, nullptr, mask, ngroups(d0),
                             wset_elems(d0, 1));
  TG_CHECK(d0->groups <= 1, TG_ENOSUP, "tg_conv2d_bwd_data(mfma): weight-set groups on a first-generation kernel");
  Geom g;   // a forward conv over gy: in = (hout,wout,cout), out = (hin,win,cin), pad' = k-1-pad
  int rc = fill_geom("tg_conv2d_bwd_data", d->n, d->hout, d->wout, d->cout, d->hin, d->win, d->cin, d->kh, d->kw,
                     d->kh - 1 - d->pad_t, d->kw - 1 - d->pad_l, &g);
  if (rc) return rc;
  if (d->kh == 1) return dispatch_fwd<1, 1>(g, (const bf16*)gy, (const bf16*)wp, nullptr, (bf16*)gx, s);
  return dispatch_fwd<3, 3>(g, (const bf16*)gy, (const bf16*)wp, nullptr, (bf16*)gx, s);
}

// Backward-data of a discriminator block's last conv straight from the gradient of the POOLED output and the layer's sign
// bytes (conv_tile.hip UNPOOL kernels): the tile kernels' shapes with 32-channel chunks of the incoming gradient
bool tg_conv2d_bwd_data_unpool_supported_mfma(const TgConvDesc* d0) {
  TgConvDesc dd;
  const TgConvDesc* d = as_dense(d0, &dd) ? &dd : d0;
  return d == d0 && is16(d) && d->algo != TG_ALGO_MFMA_V1 && d->kh == 3 && d->kw == 3 && d->cin % 8 == 0 && d->cout % 32 == 0 &&
         d->hout % 2 == 0 && d->wout % 2 == 0 &&
         tg_conv_tile_supported(d->hout, d->wout, d->hin, d->win, d->kh, d->kw, d->pad_t, d->pad_l);
}

int tg_conv2d_bwd_data_unpool_mfma(const TgConvDesc* d, const void* gy_pooled, const void* y_signs, const void* wp, void* gx,
                                   hipStream_t s, const void* mask, void* gy_out, const void* y_act) {
  TG_CHECK(tg_conv2d_bwd_data_unpool_supported_mfma(d), TG_ENOSUP, "tg_conv2d_bwd_data_unpool: not built for this layer");
  TG_CHECK(!mask || tg_conv2d_bwd_data_mask_fusable_mfma(d), TG_ENOSUP, "tg_conv2d_bwd_data_unpool: mask not fusable here");
  return tg_conv_tile_run(d->n, d->hout, d->wout, d->cout, d->cin, d->kh, d->kh - 1 - d->pad_t, 0, mask ? d->lrelu_alpha : 1.f,
                          nullptr, wp, nullptr, gx, s, mask, nullptr, 0, nullptr, nullptr, nullptr, gy_pooled, y_signs,
                          d->lrelu_alpha, gy_out, y_act, ngroups(d), wset_elems(d, 1));
}

static void wgrad_split(const Geom& g, int* n_ci, int* n_co, int* nslices, int* tiles_per_block, int* total_tiles) {
  *n_ci = (g.cin + 31) / 32;
  *n_co = (g.cout + 31) / 32;
  *total_tiles = g.tiles_x * g.tiles_y * g.tiles_img;
  int want = 768 / (*n_ci * *n_co);           // aim for one resident wave of workgroups (3 per CU at 152 VGPRs)
  if (want < 1) want = 1;
  if (want > *total_tiles) want = *total_tiles;
  *tiles_per_block = (*total_tiles + want - 1) / want;
  *nslices = (*total_tiles + *tiles_per_block - 1) / *tiles_per_block;
}

bool tg_wgrad_tile_supported(int h, int w, int hout, int wout, int kh, int kw, int pad_t, int pad_l);
size_t tg_wgrad_tile_workspace(int n, int h, int w, int cin, int cout);
int tg_wgrad_tile_run(int n, int h, int w, int cin, int cout, const void* x, const void* gy, float* gw, int accumulate,
                      void* ws, size_t ws_bytes, hipStream_t s, float* gbias = nullptr);

int tg_wgrad_slab_reduce(const float* slab, float* gw, int64_t nw, int nslices, int accumulate, hipStream_t s);

static bool use_wgrad_tile(const TgConvDesc* d) {
  return d->algo != TG_ALGO_MFMA_V1 && d->cin % 8 == 0 && d->cout % 8 == 0 &&
         tg_wgrad_tile_supported(d->hin, d->win, d->hout, d->wout, d->kh, d->kw, d->pad_t, d->pad_l);
}

size_t tg_wgrad_tile_workspace2(int na, int nb, int h, int w, int cin, int cout);
int tg_wgrad_tile_run2(int na, int nb, int h, int w, int cin, int cout, const void* xa, const void* gya, const void* xb,
                       const void* gyb, float* gw, int accumulate, void* ws, size_t ws_bytes, hipStream_t s,
                       float* gbias = nullptr, int bias_segs = 3);

// two batches (na, nb images) of one layer: supported when the tile kernel takes the layer
bool tg_conv2d_bwd_weight2_supported_mfma(const TgConvDesc* d) { return is16(d) && use_wgrad_tile(d); }
size_t tg_conv2d_bwd_weight2_workspace_mfma(const TgConvDesc* d, int nb) {
  return tg_wgrad_tile_workspace2(d->n, nb, d->hin, d->win, d->cin, d->cout);
}
int tg_conv2d_bwd_weight2_mfma(const TgConvDesc* d, int nb, const void* xa, const void* gya, const void* xb, const void* gyb,
                               float* gw, int accumulate, void* ws, size_t ws_bytes, hipStream_t s, float* gbias,
                               int bias_segs) {
  return tg_wgrad_tile_run2(d->n, nb, d->hin, d->win, d->cin, d->cout, xa, gya, xb, gyb, gw, accumulate, ws, ws_bytes, s,
                            gbias, bias_segs);
}

// does tg_conv2d_bwd_weight_mfma produce the bias gradient itself for this descriptor?
bool tg_conv2d_bwd_weight_bias_fused_mfma(const TgConvDesc* d0) {
  TgConvDesc dd;
  const TgConvDesc* d = as_dense(d0, &dd) ? &dd : d0;
  return is16(d) && use_wgrad_tile(d);
}

size_t tg_conv2d_bwd_weight_workspace_mfma(const TgConvDesc* d0) {
  TgConvDesc dd;
  const TgConvDesc* d = as_dense(d0, &dd) ? &dd : d0;
  if (use_wgrad_tile(d)) return tg_wgrad_tile_workspace(d->n, d->hin, d->win, d->cin, d->cout);
  Geom g;
  if (fill_geom("tg_conv2d_bwd_weight", d->n, d->hin, d->win, d->cin, d->hout, d->wout, d->cout, d->kh, d->kw, d->pad_t,
                d->pad_l, &g))
    return 0;
  int n_ci, n_co, nslices, tpb, total;
  wgrad_split(g, &n_ci, &n_co, &nslices, &tpb, &total);
  return (size_t)nslices * d->kh * d->kw * d->cin * d->cout * sizeof(float);
}

int tg_conv2d_bwd_weight_mfma(const TgConvDesc* d0, const void* x, const void* gy, float* gw, int accumulate, void* ws,
                              size_t ws_bytes, hipStream_t s, float* gbias) {
  TgConvDesc dd;
  const TgConvDesc* d = as_dense(d0, &dd) ? &dd : d0;
  TG_CHECK(is16(d), TG_ENOSUP, "tg_conv2d_bwd_weight(mfma): 16-bit activations only");
  TG_CHECK(!gbias || use_wgrad_tile(d), TG_ENOSUP, "tg_conv2d_bwd_weight(mfma): bias gradient not fused for this layer");
  if (use_wgrad_tile(d))
    return tg_wgrad_tile_run(d->n, d->hin, d->win, d->cin, d->cout, x, gy, gw, accumulate, ws, ws_bytes, s, gbias);
  Geom g;
  int rc = fill_geom("tg_conv2d_bwd_weight", d->n, d->hin, d->win, d->cin, d->hout, d->wout, d->cout, d->kh, d->kw,
                     d->pad_t, d->pad_l, &g);
  if (rc) return rc;
  int n_ci, n_co, nslices, tpb, total;
  wgrad_split(g, &n_ci, &n_co, &nslices, &tpb, &total);
  const int64_t nw = (int64_t)d->kh * d->kw * d->cin * d->cout;
  TG_CHECK(ws && ws_bytes >= (size_t)nslices * nw * sizeof(float), TG_EINVAL,
           "tg_conv2d_bwd_weight(mfma): workspace too small (%zu < %zu)", ws_bytes, (size_t)nslices * nw * sizeof(float));
  const int TW = 1 << g.tw_log2, TH = 1 << g.th_log2, TI = 1 << g.ti_log2;
  const int halo_px = TI * (TH + d->kh - 1) * (TW + d->kw - 1);
  TG_CHECK(halo_px * 4 <= 256 * 5, TG_ENOSUP, "conv_wgrad(mfma): halo tile too large (%d px)", halo_px);
  size_t lds = ((size_t)(halo_px * 80 + 15) & ~(size_t)15) + 128 * 80;
  if (lds < 4 * 16 * 64 * sizeof(float)) lds = 4 * 16 * 64 * sizeof(float);
  TG_CHECK(lds <= 64 * 1024, TG_ENOSUP, "conv_wgrad(mfma): LDS %zu > 64 KiB", lds);
  dim3 grid(nslices, n_ci * n_co);
  tg_note_kernel("conv_wgrad_mfma<%d,%d>", d->kh, d->kw);
  if (d->kh == 1 && tg_elem_f16())
    hipLaunchKernelGGL((conv_wgrad_mfma<1, 1, true>), grid, dim3(256), lds, s, (const bf16*)x, (const bf16*)gy, (float*)ws,
                       g, n_co, tpb, total);
  else if (d->kh == 1)
    hipLaunchKernelGGL((conv_wgrad_mfma<1, 1>), grid, dim3(256), lds, s, (const bf16*)x, (const bf16*)gy, (float*)ws, g,
                       n_co, tpb, total);
  else if (tg_elem_f16())
    hipLaunchKernelGGL((conv_wgrad_mfma<3, 3, true>), grid, dim3(256), lds, s, (const bf16*)x, (const bf16*)gy, (float*)ws,
                       g, n_co, tpb, total);
  else
    hipLaunchKernelGGL((conv_wgrad_mfma<3, 3>), grid, dim3(256), lds, s, (const bf16*)x, (const bf16*)gy, (float*)ws, g,
                       n_co, tpb, total);
  TG_LAUNCH_CHECK("conv_wgrad_mfma");
  return tg_wgrad_slab_reduce((const float*)ws, gw, nw, nslices, accumulate, s);
}
