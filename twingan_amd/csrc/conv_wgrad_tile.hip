// conv_wgrad_tile: 3x3 SAME backward-weight for feature maps of 16x16 and up, bf16 NHWC inputs,
// fp32 HWIO output.   gw[tap][ci][co] = sum over pixels of x[pix + tap][ci] * gy[pix][co]
//
// The reduction (K) dimension of this GEMM is the PIXEL axis, while both operands are stored
// channel-contiguous (NHWC): every MFMA fragment needs 8 consecutive pixels of one channel.  gfx950's
// LDS transpose read does exactly that: ds_read_b64_tr_b16 takes, per 16-lane group, sixteen 8-byte
// row addresses R[s][0..3] and returns to lane i the column R[(i>>2)+4j][i&3], j = 0..3 (measured with
// tools/probes/ds_read_tr.hip).  With lane s pointing at (pixel p0 + (s>>2), channels c0 + 4*(s&3)..)
// lane i receives channel c0+i of pixels p0..p0+3: two reads build one v_mfma_f32_32x32x16_bf16
// operand (8 pixels x 32 channels per half-wave) with no VALU work and no bank conflicts (a half-wave
// reads 4 consecutive 64-byte pixels = all 64 banks once).
//
//   workgroup = 4 waves; owns one (32 ci x 32 co) block of the weight and a contiguous range of
//   (8 rows x 16 cols) pixel tiles; per tile the x halo [10][18][32 ch] and gy [8][16][32 ch] are
//   staged in LDS (64 B per pixel, dense); wave w reduces rows 2w, 2w+1 (two 16-pixel K steps) into
//   9 per-tap 32x32 fp32 accumulators; the next tile's global loads are in flight during the MFMAs.
//   At the end the 4 waves are summed through LDS and the valid [cin x cout] part is written to this
//   workgroup's fp32 slab; conv_wgrad_slab_reduce sums the slabs into gw.
//   NW = 8 (round 3): the same wave program with EIGHT waves per workgroup on a 16-row tile (wave w: rows 2w, 2w+1):
//   one workgroup per CU carries what two did, so the launch writes and re-reads half the slabs (256 x 36 KB instead
//   of 512 x 36 KB -- the slab round trip was ~10 of a launch's 35-40 us) and a tile's halo overhead drops from
//   10*18 / (8*16) = 1.41 to 18*18 / (16*16) = 1.27 pixels staged per pixel reduced.
//
// Reference call site replaced: the Conv2DBackpropFilter gradient of tf.contrib.layers.conv2d
// (nets/pggan_utils.py:316-320).
#include "tg_common.h"

// launches conv_wgrad_tile_kernel<TW, BIAS, F16, NW> in the element format of the current call (tg_elem_f16)
#define TG_WG_LAUNCH(TW_, BIAS_, NW_, ...)                                                        \
  do {                                                                                            \
    if (tg_elem_f16()) hipLaunchKernelGGL((conv_wgrad_tile_kernel<TW_, BIAS_, true, NW_>), __VA_ARGS__); \
    else hipLaunchKernelGGL((conv_wgrad_tile_kernel<TW_, BIAS_, false, NW_>), __VA_ARGS__);         \
  } while (0)
#include <cstdlib>
#include <mutex>
#include <type_traits>

namespace {

struct WgGeom {
  int n, h, w, cin, cout;
  int tiles_x, tiles_y, total_tiles;
  int n_co_blk;                 // number of 32-wide co blocks (pair = ci_blk * n_co_blk + co_blk)
  int n_pairs, nslices;
  int tiles_per_wg;
  // fused input concat(nearest_up2(x), x1): x is [n, h/2, w/2, c0], x1 is [n1, h, w, cin - c0] (c0 = 0: plain input);
  // image i reads skip image perm-group(i).  A 32-channel ci block lies entirely in one source (c0 % 32 == 0).
  const bf16* x1;
  int c0, gsz;
  unsigned perm;
  // second (x, gy) segment of the SAME layer (another batch whose gradient goes to the same weight): tiles
  // [tiles_a, total_tiles) read it -- one launch instead of two (every launch costs ~15-20 us of ramp, slab write and
  // reduction whatever its size).  tiles_a == total_tiles: no second segment.
  const bf16* xb;
  const bf16* gyb;
  int nb, tiles_a;
  // BIAS kernels: gbias[co] += sum over all pixels of gy[.., co] (BiasAddGrad of the layer), by the ci-block-0
  // workgroups: one more MFMA per K step with an all-ones A operand.
  float* gbias;
  int bias_segs;      // bit 0: segment a contributes to gbias, bit 1: segment b
  int nw;             // waves per workgroup of the tile kernel: 4 (8-row tiles) or 8 (16-row tiles)
  int quad;           // 1: conv_wgrad_quad_kernel (64 x 64 blocks, 8 waves); n_co_blk / n_pairs then count 64-wide blocks
  int vec_ok;         // 16-byte slab stores: cout % 4 == 0 and the slab 16-byte aligned (set by the launchers)
};

extern __shared__ __attribute__((aligned(16))) unsigned char wg_smem[];

typedef __attribute__((ext_vector_type(4))) short s16x4;
constexpr unsigned WOOB = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t wg_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// 8 consecutive pixels (stride 64 B) of this lane's channel: two transpose reads
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* p) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * 64));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
  v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(bf16x8, v);
}

// TW = 16: a tile is 8 rows x 16 cols of one image (maps of 16x16 and up).  TW = 8: the 8x8 maps -- a tile is TWO
// whole images, K step ks = image ks of the pair, and the two 8-pixel halves of a K step are rows 2w and 2w+1.
template <int TW, bool BIAS = false, bool F16 = false, int NW = 4>
__global__ __launch_bounds__(64 * NW) void conv_wgrad_tile_kernel(const bf16* __restrict__ x, const bf16* __restrict__ gy,
                                                                 float* __restrict__ slab, const WgGeom g) {
  static_assert(NW == 4 || (NW == 8 && TW == 16), "8 waves: 16-row tiles of the 16-column kernel only");
  constexpr int THREADS = 64 * NW;
  constexpr int TH = 2 * NW, HWX = TW + 2, HH = TH + 2, NT = 9;        // TW = 16: wave w reduces rows 2w, 2w + 1
  constexpr int IPT = 16 / TW;                      // images per tile
  constexpr int PS = 64;                            // LDS bytes per pixel (32 channels, dense)
  constexpr int XPX = IPT * HH * HWX;               // halo pixels per tile: 180 / 200 (NW = 4), 324 (NW = 8)
  constexpr int XVEC = XPX * 4, XSLOTS = (XVEC + THREADS - 1) / THREADS;   // 720 -> 3, 800 -> 4, 1296 / 512 -> 3
  constexpr int GPX = TW == 16 ? TH * 16 : 128;                        // output pixels per tile: 128 / 256
  constexpr int GSLOTS = (GPX * 4) / THREADS;                          // 2
  constexpr int X_BYTES = XPX * PS;                                    // 11520 / 12800 / 20736
  constexpr int G_BYTES = GPX * PS;                                    // 8192 / 16384

  unsigned char* sX = wg_smem;
  unsigned char* sG = wg_smem + X_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  // 1-D grid over (pixel slice, channel pair).  A 64-byte half of a 128-byte x line belongs to ONE ci block, so
  // the two ci blocks that share a line must read it at the same time on the same XCD (same L2) or HBM serves
  // it twice (PMC: 1.9x the algorithmic bytes before this mapping).  Workgroup b runs on XCD b % 8, hence
  // id = (slice / 8) * 8 * npairs + pair * 8 + slice % 8.
  int slice, pair;
  {
    const int id = blockIdx.x, npairs = g.n_pairs;
    if ((g.nslices & 7) == 0) {
      const int hi = id / (8 * npairs), rem = id - hi * 8 * npairs;
      pair = rem >> 3;
      slice = hi * 8 + (rem & 7);
    } else {
      pair = id % npairs;
      slice = id / npairs;
    }
  }
  const int ci_blk = pair / g.n_co_blk, co_blk = pair - ci_blk * g.n_co_blk;
  const int ci0 = ci_blk * 32, co0 = co_blk * 32;

  // ---- staging slots: everything that does not depend on the tile is folded into per-slot constants, so a
  // tile's 5 global addresses cost one add, two range compares and a select each (written branch-free: left
  // to itself the compiler branches around every load and drains vmcnt inside the branch).
  int x_loff[XSLOTS], x_hy1[XSLOTS], x_hx1[XSLOTS], x_rel[XSLOTS];
  bool x_use[XSLOTS];
#pragma unroll
  for (int s = 0; s < XSLOTS; ++s) {
    const int v = tid + s * THREADS;
    const int px = v >> 2, part = v & 3;
    const int sub = px / (HH * HWX), rem = px - sub * (HH * HWX);          // image of the tile (TW = 8), pixel in its halo
    x_hy1[s] = rem / HWX - 1;
    x_hx1[s] = rem % HWX - 1;
    x_use[s] = v < XVEC && ci0 + part * 8 + 8 <= g.cin;                    // else zero fill
    // bytes from the tile's first pixel (TW = 8: from the first pixel of the pair's first image)
    if (g.c0 == 0)
      x_rel[s] = (((sub * g.h + x_hy1[s]) * g.w + x_hx1[s]) * g.cin + ci0 + part * 8) * 2;
    else if (ci0 < g.c0)      // half-resolution source: (iy >> 1, ix >> 1); tile origins are even
      x_rel[s] = (((x_hy1[s] >> 1) * (g.w >> 1) + (x_hx1[s] >> 1)) * g.c0 + ci0 + part * 8) * 2;
    else                      // skip source
      x_rel[s] = ((x_hy1[s] * g.w + x_hx1[s]) * (g.cin - g.c0) + ci0 - g.c0 + part * 8) * 2;
    x_loff[s] = px * PS + part * 16;
  }
  int g_loff[GSLOTS];
  unsigned g_rel[GSLOTS];
#pragma unroll
  for (int s = 0; s < GSLOTS; ++s) {
    const int v = tid + s * THREADS;
    const int px = v >> 2, part = v & 3;
    const bool use = co0 + part * 8 + 8 <= g.cout;
    // TW = 16: pixel (px >> 4, px & 15) of the tile; TW = 8: pixel px & 63 of image px >> 6 (rows are contiguous)
    g_rel[s] = use ? (unsigned)(((TW == 16 ? (px >> 4) * g.w + (px & 15) : px) * g.cout + co0 + part * 8) * 2) : WOOB;
    g_loff[s] = px * PS + part * 16;
  }

  // ---- fragment addresses.  16-lane group gq = lane >> 4: channel half c0 = 16*(gq & 1), K half
  // kg = gq >> 1 (pixels kg*8 .. kg*8+7 of the 16-pixel K step); lane t = lane & 15 points at
  // pixel +(t >> 2), channels c0 + 4*(t & 3).
  const int gq = lane >> 4, t16 = lane & 15;
  const int frag_off = ((gq >> 1) * 8 + (t16 >> 2)) * PS + ((gq & 1) * 16 + (t16 & 3) * 4) * 2;
  // x: the second 8-pixel half of a K step is 8 pixels on (TW = 16) or the next halo row (TW = 8)
  const int frag_off_x = ((gq >> 1) * (TW == 16 ? 8 : HWX) + (t16 >> 2)) * PS + ((gq & 1) * 16 + (t16 & 3) * 4) * 2;
  // TW = 16: wave wid reduces tile rows 2*wid (K step 0) and 2*wid + 1 (K step 1)

  f32x16 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  f32x16 accb;                 // BIAS: every row = the column sums of gy
  bf16x8 ones;
  if constexpr (BIAS) {
#pragma unroll
    for (int j = 0; j < 16; ++j) accb[j] = 0.f;
    {      // 1.0 in the element format, eight times
      typedef __attribute__((ext_vector_type(4))) unsigned u4;
      u4 o4;
      o4[0] = o4[1] = o4[2] = o4[3] = ones16x2<F16>();
      ones = __builtin_bit_cast(bf16x8, o4);
    }
  }
  const bool do_bias = BIAS && ci_blk == 0;      // uniform
  bool bias_a = false, bias_b = false, bias_0 = false, bias_1 = false;      // per register stage / per LDS buffer

  const bool from_up = g.c0 != 0 && ci0 < g.c0, from_skip = g.c0 != 0 && ci0 >= g.c0;
  const int xc = from_up ? g.c0 : (from_skip ? g.cin - g.c0 : g.cin);      // channels per pixel of this block's source
  const size_t ximg = from_up ? (size_t)(g.h >> 1) * (g.w >> 1) * xc : (size_t)g.h * g.w * xc;
  const size_t gimg = (size_t)g.h * g.w * g.cout;
  const bf16* xsrc = from_skip ? g.x1 : x;
  const int tile_begin = slice * g.tiles_per_wg;
  int tile_end = tile_begin + g.tiles_per_wg;
  if (tile_end > g.total_tiles) tile_end = g.total_tiles;

  // Two register stages (tiles t+1 and t+2 in flight while tile t is reduced) and two LDS buffers (one
  // barrier per tile: tile t+1 is written to the other buffer while slower waves still read tile t).
  // With a single stage the loads had only one tile's MFMAs (~0.4 us) to land and every tile exposed
  // most of an HBM round trip.
  struct Stage {
    bf16x8 rx[XSLOTS], rg[GSLOTS];
  };
  // Position of the next tile to request.  Tiles are requested strictly in order, so (tile column, tile row, image) and
  // the skip source's permuted image advance by compare-and-carry; the divisions run once per workgroup (and once more
  // where the second segment starts).  The scalar unit issued 250 instructions per loop trip for the divisions before.
  struct Cursor {
    int tile, tx, ty, img, grp, rem;      // grp / rem: img = grp * gsz + rem (skip source with a group permutation)
  };
  const bool permuted = from_skip && g.gsz != 0;
  auto cursor_set = [&](Cursor& c, int tile) __attribute__((always_inline)) {
    c.tile = tile;
    int t = tile >= g.tiles_a ? tile - g.tiles_a : tile;
    if constexpr (TW == 16) {
      c.tx = t % g.tiles_x;
      t /= g.tiles_x;
      c.ty = t % g.tiles_y;
      c.img = t / g.tiles_y;
    } else {
      c.tx = c.ty = 0;
      c.img = t * 2;      // the pair (2t, 2t+1)
    }
    c.grp = permuted ? c.img / g.gsz : 0;
    c.rem = permuted ? c.img - c.grp * g.gsz : 0;
  };
  auto cursor_next = [&](Cursor& c) __attribute__((always_inline)) {
    ++c.tile;
    if (c.tile == g.tiles_a || (TW != 16 && permuted)) {      // segment switch / 8x8 pairs of a permuted source: recompute
      cursor_set(c, c.tile);
      return;
    }
    if constexpr (TW == 16) {
      if (++c.tx == g.tiles_x) {
        c.tx = 0;
        if (++c.ty == g.tiles_y) {
          c.ty = 0;
          ++c.img;
          if (permuted && ++c.rem == g.gsz) {
            c.rem = 0;
            ++c.grp;
          }
        }
      }
    } else {
      c.img += 2;
    }
  };
  auto load_tile = [&](Stage& st, Cursor& c, bool& bias_on) __attribute__((always_inline)) {
    const unsigned live = c.tile < tile_end;      // past the end: every lane out of range -> a tile of zeros
    const bool segb = c.tile >= g.tiles_a;        // second (x, gy) pair
    bias_on = do_bias && ((g.bias_segs >> (segb ? 1 : 0)) & 1);
    const bf16* xs = segb ? g.xb : xsrc;
    const bf16* gs = segb ? g.gyb : gy;
    const int nseg = segb ? g.nb : g.n;
    const int img = c.img, ox0 = c.tx * TW, oy0 = c.ty * TH;
    // TW = 8: an odd batch ends with a half-empty tile: the buffer resource then covers one image, the other reads zeros
    const int nimg = TW == 16 ? 1 : (img + 1 < nseg ? 2 : 1);
    const int ximg_i = permuted ? (int)((g.perm >> (8 * c.grp)) & 0xffu) * g.gsz + c.rem : img;
    const __amdgpu_buffer_rsrc_t bx = wg_rsrc(xs + (size_t)ximg_i * ximg, (unsigned)(ximg * 2 * nimg));
    const __amdgpu_buffer_rsrc_t bg = wg_rsrc(gs + (size_t)img * gimg, (unsigned)(gimg * 2 * nimg));
    const int xbase = from_up ? ((oy0 >> 1) * (g.w >> 1) + (ox0 >> 1)) * xc * 2 : (oy0 * g.w + ox0) * xc * 2;
    const unsigned gbase = (unsigned)((oy0 * g.w + ox0) * g.cout * 2);
#pragma unroll
    for (int s = 0; s < XSLOTS; ++s) {
      const unsigned ok = live & (unsigned)x_use[s] & (unsigned)((unsigned)(oy0 + x_hy1[s]) < (unsigned)g.h) &
                          (unsigned)((unsigned)(ox0 + x_hx1[s]) < (unsigned)g.w);
      const unsigned off = ok ? (unsigned)(xbase + x_rel[s]) : WOOB;
      st.rx[s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(bx, off, 0, 0));
    }
#pragma unroll
    for (int s = 0; s < GSLOTS; ++s)      // g_rel = WOOB (bit 31) stays out of range after adding gbase < 2^31
      st.rg[s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(bg, live ? gbase + g_rel[s] : WOOB, 0, 0));
    cursor_next(c);
  };
  auto stage_to_lds = [&](const Stage& st, unsigned char* bX, unsigned char* bG) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < XSLOTS; ++s)
      if (s < XSLOTS - 1 || tid + s * THREADS < XVEC) *reinterpret_cast<bf16x8*>(bX + x_loff[s]) = st.rx[s];
#pragma unroll
    for (int s = 0; s < GSLOTS; ++s) *reinterpret_cast<bf16x8*>(bG + g_loff[s]) = st.rg[s];
  };
  // One tile = 2 K steps x 9 taps = 18 (+2 bias) MFMAs.  TW = 16: K step 1 is the next row of the same image, so its
  // taps (ky, kx) read the halo fragments of K step 0's taps (ky + 1, kx): 4 halo rows x 3 columns = 12 x fragments + 2
  // gy fragments = 28 transpose reads per tile.  Left to the scheduler the reads came in bursts of 4-8 right before their
  // MFMAs (an LDS round trip exposed per burst: ~1000 of a tile's ~3400 cycles).  Here they are requested in consumption
  // order, five fragments ahead of the MFMAs, and the interleaving is pinned with scheduling-group barriers: every read
  // has >= 3 MFMAs (~100 cycles) to land.
  // BIAS kernels: the bias MFMAs are issued by every workgroup (A operand = ones where this tile feeds the bias gradient,
  // zeros elsewhere -- a branch would split the scheduling region, and a second copy of the tile code for the
  // ci-block-0 workgroups doubled the accumulator registers: 256 AGPRs, one wave per SIMD, +45 % time).
  auto reduce_tile = [&](const unsigned char* bX, const unsigned char* bG, bool bias_on) __attribute__((always_inline)) {
    constexpr bool WB = BIAS;
    bf16x8 bo;
    if constexpr (WB) {
      typedef __attribute__((ext_vector_type(4))) unsigned u4;
      const u4 o = __builtin_bit_cast(u4, ones);
      u4 sel;
#pragma unroll
      for (int j = 0; j < 4; ++j) sel[j] = bias_on ? o[j] : 0u;
      bo = __builtin_bit_cast(bf16x8, sel);
    }
    auto gaddr = [&](int ks) { return bG + (TW == 16 ? (wid * 2 + ks) * 16 : ks * 64 + wid * 16) * PS + frag_off; };
    if constexpr (TW == 16) {
      // X[row][kx]: halo row 2 wid + row, column offset kx
      auto xaddr = [&](int row, int kx) { return bX + ((wid * 2 + row) * HWX + kx) * PS + frag_off_x; };
      bf16x8 g0, g1, X[4][3];
      g0 = tr_frag(gaddr(0));
      g1 = tr_frag(gaddr(1));
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) X[0][kx] = tr_frag(xaddr(0, kx));
      if constexpr (WB) {
        accb = mfma_32x32x16<F16>(bo, g0, accb);
        accb = mfma_32x32x16<F16>(bo, g1, accb);
      }
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {      // halo row 0: K step 0, ky = 0
        acc[kx] = mfma_32x32x16<F16>(X[0][kx], g0, acc[kx]);
        X[1][kx] = tr_frag(xaddr(1, kx));
      }
#pragma unroll
      for (int row = 1; row < 3; ++row)     // halo rows 1, 2: K step 0 tap row `row`, K step 1 tap row `row - 1`
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          acc[row * 3 + kx] = mfma_32x32x16<F16>(X[row][kx], g0, acc[row * 3 + kx]);
          acc[(row - 1) * 3 + kx] = mfma_32x32x16<F16>(X[row][kx], g1, acc[(row - 1) * 3 + kx]);
          X[row + 1][kx] = tr_frag(xaddr(row + 1, kx));
        }
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) acc[6 + kx] = mfma_32x32x16<F16>(X[3][kx], g1, acc[6 + kx]);      // halo row 3
      __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);
      if constexpr (WB) __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
    } else {
      // TW = 8: K step ks = image ks of the pair -- nothing shared: 2 x (gy + 9 x) fragments
      auto xaddr = [&](int ks, int tap) {
        return bX + (ks * (HH * HWX) + wid * 2 * HWX + (tap / 3) * HWX + tap % 3) * PS + frag_off_x;
      };
      bf16x8 gf[2], xf[2][NT];
      gf[0] = tr_frag(gaddr(0));
#pragma unroll
      for (int tap = 0; tap < 3; ++tap) xf[0][tap] = tr_frag(xaddr(0, tap));
#pragma unroll
      for (int tap = 0; tap < NT; ++tap) {
        acc[tap] = mfma_32x32x16<F16>(xf[0][tap], gf[0], acc[tap]);
        if (tap + 3 < NT) xf[0][tap + 3] = tr_frag(xaddr(0, tap + 3));
        else if (tap + 3 == NT) gf[1] = tr_frag(gaddr(1));
        else xf[1][tap + 2 - NT] = tr_frag(xaddr(1, tap + 2 - NT));      // taps 7, 8 -> x1[0], x1[1]
      }
#pragma unroll
      for (int tap = 0; tap < NT; ++tap) {
        acc[tap] = mfma_32x32x16<F16>(xf[1][tap], gf[1], acc[tap]);
        if (tap + 2 < NT) xf[1][tap + 2] = tr_frag(xaddr(1, tap + 2));
      }
      if constexpr (WB) {
        accb = mfma_32x32x16<F16>(bo, gf[0], accb);
        accb = mfma_32x32x16<F16>(bo, gf[1], accb);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 2 + (WB ? 2 : 0), 0);
    }
  };

  unsigned char* sX1 = wg_smem + X_BYTES + G_BYTES;
  unsigned char* sG1 = sX1 + X_BYTES;
  // Two tiles per trip, no branches inside: a workgroup with an odd tile count reduces one all-zero tile.
  Stage sa, sb;
  Cursor cur;
  cursor_set(cur, tile_begin);
  load_tile(sa, cur, bias_a);
  load_tile(sb, cur, bias_b);
  for (int tile = tile_begin; tile < tile_end; tile += 2) {
    stage_to_lds(sa, sX, sG);
    bias_0 = bias_a;
    __syncthreads();
    load_tile(sa, cur, bias_a);
    reduce_tile(sX, sG, bias_0);
    stage_to_lds(sb, sX1, sG1);
    bias_1 = bias_b;
    __syncthreads();
    load_tile(sb, cur, bias_b);
    reduce_tile(sX1, sG1, bias_1);
  }

  // ---- cross-wave reduction through LDS and the slab write.
  // acc[tap][r]: ci = ci0 + (r & 3) + 8*(r >> 2) + 4*(lane >> 5), co = co0 + (lane & 31)
  // A thread owns FOUR consecutive co of one ci: 16-byte LDS reads of each wave's partial sums, one 16-byte slab store.
  // TPP taps per phase (two with 8 waves, so that all 512 threads hold a float4): 18 barriers -> 10.  The
  // one-element-per-thread form (4-byte stores of 128-byte runs, two barriers per tap) cost 8 us of a 15-19 us launch at
  // n = 16 -- 5 us of it the stores (profiles/r05_c_wgrad_epilogue_probes.txt); the sums are taken in the same order.
  constexpr int TPP = NW / 4;
  float* red = reinterpret_cast<float*>(wg_smem);    // [TPP][NW waves][16 regs][64 lanes] floats = 16 / 64 KiB
  float* out = slab + (size_t)slice * NT * g.cin * g.cout;
  {
    const int tp = tid >> 8, q = tid & 255;
    const int r = q >> 4, l4 = (q & 15) * 4;
    const int ci = ci0 + (r & 3) + 8 * (r >> 2) + 4 * (l4 >> 5);
    const int co = co0 + (l4 & 31);
#pragma unroll
    for (int p = 0; p < (NT + TPP - 1) / TPP; ++p) {
      __syncthreads();
#pragma unroll
      for (int t = 0; t < TPP; ++t) {
        if (p * TPP + t < NT) {
#pragma unroll
          for (int rr = 0; rr < 16; ++rr) red[((t * NW + wid) * 16 + rr) * 64 + lane] = acc[p * TPP + t][rr];
        }
      }
      __syncthreads();
      const int tap = p * TPP + tp;
      if (tap < NT && ci < g.cin && co < g.cout) {
        const float* src = red + ((tp * NW) * 16 + r) * 64 + l4;
        auto part = [&](int w) { return *reinterpret_cast<const f32x4*>(src + w * 16 * 64); };
        f32x4 sum = part(0) + part(1) + part(2) + part(3);      // waves in a fixed order
        if constexpr (NW == 8) sum += part(4) + part(5) + part(6) + part(7);
        float* dst = out + ((size_t)tap * g.cin + ci) * g.cout + co;
        if (g.vec_ok) {      // cout % 4 == 0 and a 16-byte aligned slab
          *reinterpret_cast<f32x4*>(dst) = sum;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (co + j < g.cout) dst[j] = sum[j];
        }
      }
    }
  }
  if constexpr (BIAS) {
    if (do_bias) {      // accb[0] on lanes 0..31 = row 0 of the all-equal rows: column sum of co0 + lane
      __syncthreads();
      if (lane < 32) red[wid * 32 + lane] = accb[0];
      __syncthreads();
      if (tid < 32 && co0 + tid < g.cout) {
        float t = red[tid] + red[32 + tid] + red[64 + tid] + red[96 + tid];
        if constexpr (NW == 8) t += red[128 + tid] + red[160 + tid] + red[192 + tid] + red[224 + tid];
        atomicAdd(g.gbias + co0 + tid, t);
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------------------------
// Quadrant kernel (round 3): one workgroup owns a 64 ci x 64 co block of the filter gradient -- FOUR 32 x 32 quadrants,
// each reduced by two waves (the two 4-row halves of an 8 x 16 tile), 8 waves in all.  Why (same per-wave program as
// conv_wgrad_tile_kernel, different block shape):
//   * a workgroup stages 64 channels of x and of gy = WHOLE 128-byte lines (a 32-channel block of a 64-channel tensor
//     reads one half of every line; its neighbour block, on another CU, the other half: each line crossed L2 -> L1 twice);
//   * 39 KB staged per tile feed 8 x 36 = 288 MFMAs instead of 19.7 KB for 72: half the L2 -> LDS bytes per MFMA, and a
//     quadrant's operands are shared by two waves instead of being re-fetched by another workgroup;
//   * 36 MFMAs per wave between two barriers instead of 18.
// LDS holds the two 32-channel halves of each operand as separate PLANES (64 B per pixel, dense) so that the transpose
// reads keep the conflict-free pattern of the tile kernel.  cin % 64 == 0 and cout % 64 == 0 (a concat input: c0 % 64 == 0).
template <bool BIAS = false, bool F16 = false>
__global__ __launch_bounds__(512) void conv_wgrad_quad_kernel(const bf16* __restrict__ x, const bf16* __restrict__ gy,
                                                              float* __restrict__ slab, const WgGeom g) {
  constexpr int THREADS = 512, TW = 16, TH = 8, HWX = 18, HH = 10, NT = 9, RPW = 4;
  constexpr int PS = 64;                                     // LDS bytes per pixel of one plane (32 channels)
  constexpr int XPX = HH * HWX, GPX = TH * TW;               // 180 halo pixels, 128 output pixels
  constexpr int XPLANE = XPX * PS, GPLANE = GPX * PS;        // 11520, 8192
  constexpr int XVEC = XPX * 8, XSLOTS = (XVEC + THREADS - 1) / THREADS;      // 1440 -> 3
  constexpr int GSLOTS = (GPX * 8) / THREADS;                                 // 2
  constexpr int X_BYTES = 2 * XPLANE, G_BYTES = 2 * GPLANE;                   // 23040 + 16384 per buffer

  unsigned char* sX = wg_smem;
  unsigned char* sG = wg_smem + X_BYTES;
  unsigned char* sX1 = wg_smem + X_BYTES + G_BYTES;
  unsigned char* sG1 = sX1 + X_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int qi = wid & 1, qo = (wid >> 1) & 1, kh = wid >> 2;      // ci half, co half, 4-row half of the tile
  int slice, pair;
  {      // slices of one pair on consecutive XCDs, the pairs of a slice on the same XCD (see conv_wgrad_tile_kernel)
    const int id = blockIdx.x, npairs = g.n_pairs;
    if ((g.nslices & 7) == 0) {
      const int hi = id / (8 * npairs), rem = id - hi * 8 * npairs;
      pair = rem >> 3;
      slice = hi * 8 + (rem & 7);
    } else {
      pair = id % npairs;
      slice = id / npairs;
    }
  }
  const int ci_blk = pair / g.n_co_blk, co_blk = pair - ci_blk * g.n_co_blk;
  const int ci0 = ci_blk * 64, co0 = co_blk * 64;

  // ---- staging slots: 8 lanes fetch the 8 x 16 B of one pixel's 64 channels (one 128-byte line)
  int x_loff[XSLOTS], x_hy1[XSLOTS], x_hx1[XSLOTS], x_rel[XSLOTS];
  bool x_use[XSLOTS];
#pragma unroll
  for (int s = 0; s < XSLOTS; ++s) {
    const int v = tid + s * THREADS;
    const int px = v >> 3, part = v & 7;
    x_hy1[s] = px / HWX - 1;
    x_hx1[s] = px % HWX - 1;
    x_use[s] = v < XVEC && ci0 + part * 8 + 8 <= g.cin;
    if (g.c0 == 0)
      x_rel[s] = ((x_hy1[s] * g.w + x_hx1[s]) * g.cin + ci0 + part * 8) * 2;
    else if (ci0 < g.c0)      // half-resolution source: (iy >> 1, ix >> 1); tile origins are even
      x_rel[s] = (((x_hy1[s] >> 1) * (g.w >> 1) + (x_hx1[s] >> 1)) * g.c0 + ci0 + part * 8) * 2;
    else                      // skip source
      x_rel[s] = ((x_hy1[s] * g.w + x_hx1[s]) * (g.cin - g.c0) + ci0 - g.c0 + part * 8) * 2;
    x_loff[s] = (part >> 2) * XPLANE + px * PS + (part & 3) * 16;
  }
  int g_loff[GSLOTS];
  unsigned g_rel[GSLOTS];
#pragma unroll
  for (int s = 0; s < GSLOTS; ++s) {
    const int v = tid + s * THREADS;
    const int px = v >> 3, part = v & 7;
    const bool use = co0 + part * 8 + 8 <= g.cout;
    g_rel[s] = use ? (unsigned)((((px >> 4) * g.w + (px & 15)) * g.cout + co0 + part * 8) * 2) : WOOB;
    g_loff[s] = (part >> 2) * GPLANE + px * PS + (part & 3) * 16;
  }

  // ---- fragment addresses inside a plane (as conv_wgrad_tile_kernel, TW = 16)
  const int gq = lane >> 4, t16 = lane & 15;
  const int frag_off = ((gq >> 1) * 8 + (t16 >> 2)) * PS + ((gq & 1) * 16 + (t16 & 3) * 4) * 2;

  f32x16 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  f32x16 accb;
  bf16x8 ones;
  if constexpr (BIAS) {
#pragma unroll
    for (int j = 0; j < 16; ++j) accb[j] = 0.f;
    typedef __attribute__((ext_vector_type(4))) unsigned u4;
    u4 o4;
    o4[0] = o4[1] = o4[2] = o4[3] = ones16x2<F16>();
    ones = __builtin_bit_cast(bf16x8, o4);
  }
  const bool do_bias = BIAS && ci_blk == 0 && qi == 0;      // wave-uniform: the ci-half-0 waves of the ci-block-0 workgroups
  bool bias_a = false, bias_b = false, bias_0 = false, bias_1 = false;

  const bool from_up = g.c0 != 0 && ci0 < g.c0, from_skip = g.c0 != 0 && ci0 >= g.c0;
  const int xc = from_up ? g.c0 : (from_skip ? g.cin - g.c0 : g.cin);
  const size_t ximg = from_up ? (size_t)(g.h >> 1) * (g.w >> 1) * xc : (size_t)g.h * g.w * xc;
  const size_t gimg = (size_t)g.h * g.w * g.cout;
  const bf16* xsrc = from_skip ? g.x1 : x;
  const int tile_begin = slice * g.tiles_per_wg;
  int tile_end = tile_begin + g.tiles_per_wg;
  if (tile_end > g.total_tiles) tile_end = g.total_tiles;

  struct Stage {
    bf16x8 rx[XSLOTS], rg[GSLOTS];
  };
  struct Cursor {
    int tile, tx, ty, img, grp, rem;
  };
  const bool permuted = from_skip && g.gsz != 0;
  auto cursor_set = [&](Cursor& c, int tile) __attribute__((always_inline)) {
    c.tile = tile;
    int t = tile >= g.tiles_a ? tile - g.tiles_a : tile;
    c.tx = t % g.tiles_x;
    t /= g.tiles_x;
    c.ty = t % g.tiles_y;
    c.img = t / g.tiles_y;
    c.grp = permuted ? c.img / g.gsz : 0;
    c.rem = permuted ? c.img - c.grp * g.gsz : 0;
  };
  auto cursor_next = [&](Cursor& c) __attribute__((always_inline)) {
    ++c.tile;
    if (c.tile == g.tiles_a) {      // the second (x, gy) segment starts: recompute
      cursor_set(c, c.tile);
      return;
    }
    if (++c.tx == g.tiles_x) {
      c.tx = 0;
      if (++c.ty == g.tiles_y) {
        c.ty = 0;
        ++c.img;
        if (permuted && ++c.rem == g.gsz) {
          c.rem = 0;
          ++c.grp;
        }
      }
    }
  };
  auto load_tile = [&](Stage& st, Cursor& c, bool& bias_on) __attribute__((always_inline)) {
    const unsigned live = c.tile < tile_end;      // past the end: every lane out of range -> a tile of zeros
    const bool segb = c.tile >= g.tiles_a;
    bias_on = do_bias && ((g.bias_segs >> (segb ? 1 : 0)) & 1);
    const bf16* xs = segb ? g.xb : xsrc;
    const bf16* gs = segb ? g.gyb : gy;
    const int img = c.img, ox0 = c.tx * TW, oy0 = c.ty * TH;
    const int ximg_i = permuted ? (int)((g.perm >> (8 * c.grp)) & 0xffu) * g.gsz + c.rem : img;
    const __amdgpu_buffer_rsrc_t bx = wg_rsrc(xs + (size_t)ximg_i * ximg, (unsigned)(ximg * 2));
    const __amdgpu_buffer_rsrc_t bg = wg_rsrc(gs + (size_t)img * gimg, (unsigned)(gimg * 2));
    const int xbase = from_up ? ((oy0 >> 1) * (g.w >> 1) + (ox0 >> 1)) * xc * 2 : (oy0 * g.w + ox0) * xc * 2;
    const unsigned gbase = (unsigned)((oy0 * g.w + ox0) * g.cout * 2);
#pragma unroll
    for (int s = 0; s < XSLOTS; ++s) {
      const unsigned ok = live & (unsigned)x_use[s] & (unsigned)((unsigned)(oy0 + x_hy1[s]) < (unsigned)g.h) &
                          (unsigned)((unsigned)(ox0 + x_hx1[s]) < (unsigned)g.w);
      const unsigned off = ok ? (unsigned)(xbase + x_rel[s]) : WOOB;
      st.rx[s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(bx, off, 0, 0));
    }
#pragma unroll
    for (int s = 0; s < GSLOTS; ++s)
      st.rg[s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(bg, live ? gbase + g_rel[s] : WOOB, 0, 0));
    cursor_next(c);
  };
  auto stage_to_lds = [&](const Stage& st, unsigned char* bX, unsigned char* bG) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < XSLOTS; ++s)
      if (s < XSLOTS - 1 || tid + s * THREADS < XVEC) *reinterpret_cast<bf16x8*>(bX + x_loff[s]) = st.rx[s];
#pragma unroll
    for (int s = 0; s < GSLOTS; ++s) *reinterpret_cast<bf16x8*>(bG + g_loff[s]) = st.rg[s];
  };
  // One tile, one wave: rows r = 0..3 of its half (K steps), halo rows h = 0..5.  Halo row h serves tap row ky of
  // output row r = h - ky: 3 x fragments per halo row, each used by up to 3 MFMAs per tap column -- 18 x + 4 gy fragments
  // (44 transpose reads) for 36 MFMAs.  The x fragments of row h + 1 are requested before the MFMAs of row h.
  auto reduce_tile = [&](const unsigned char* bX, const unsigned char* bG, bool bias_on) __attribute__((always_inline)) {
    const unsigned char* xb = bX + qi * XPLANE + frag_off;
    const unsigned char* gb = bG + qo * GPLANE + frag_off;
    bf16x8 G[RPW], X[3], Xn[3];
#pragma unroll
    for (int r = 0; r < RPW; ++r) G[r] = tr_frag(gb + ((kh * RPW + r) * TW) * PS);
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) X[kx] = tr_frag(xb + ((kh * RPW) * HWX + kx) * PS);
    if constexpr (BIAS) {
      typedef __attribute__((ext_vector_type(4))) unsigned u4;
      const u4 o = __builtin_bit_cast(u4, ones);
      u4 sel;
#pragma unroll
      for (int j = 0; j < 4; ++j) sel[j] = bias_on ? o[j] : 0u;
      const bf16x8 bo = __builtin_bit_cast(bf16x8, sel);
#pragma unroll
      for (int r = 0; r < RPW; ++r) accb = mfma_32x32x16<F16>(bo, G[r], accb);
    }
#pragma unroll
    for (int h = 0; h < RPW + 2; ++h) {
      if (h + 1 < RPW + 2) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) Xn[kx] = tr_frag(xb + ((kh * RPW + h + 1) * HWX + kx) * PS);
      }
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int r = h - ky;
        if (r >= 0 && r < RPW) {
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] = mfma_32x32x16<F16>(X[kx], G[r], acc[ky * 3 + kx]);
        }
      }
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) X[kx] = Xn[kx];
    }
  };

  Stage sa, sb;
  Cursor cur;
  cursor_set(cur, tile_begin);
  load_tile(sa, cur, bias_a);
  load_tile(sb, cur, bias_b);
  for (int tile = tile_begin; tile < tile_end; tile += 2) {      // two tiles per trip; an odd count reduces one zero tile
    stage_to_lds(sa, sX, sG);
    bias_0 = bias_a;
    __syncthreads();
    load_tile(sa, cur, bias_a);
    reduce_tile(sX, sG, bias_0);
    stage_to_lds(sb, sX1, sG1);
    bias_1 = bias_b;
    __syncthreads();
    load_tile(sb, cur, bias_b);
    reduce_tile(sX1, sG1, bias_1);
  }

  // ---- the two 4-row halves of every quadrant are summed through LDS (kh = 0 first: a fixed order), two taps per phase;
  // a thread owns four consecutive co of one ci (16-byte LDS reads, 16-byte slab stores: see conv_wgrad_tile_kernel)
  // acc[tap][r]: ci = ci0 + 32 qi + (r & 3) + 8 (r >> 2) + 4 (lane >> 5), co = co0 + 32 qo + (lane & 31)
  float* red = reinterpret_cast<float*>(wg_smem);      // [2 taps][8 waves][16 regs][64 lanes] floats = 64 KiB
  float* out = slab + (size_t)slice * NT * g.cin * g.cout;
#pragma unroll
  for (int p = 0; p < (NT + 1) / 2; ++p) {
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (p * 2 + t < NT) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((t * 8 + wid) * 16 + r) * 64 + lane] = acc[p * 2 + t][r];
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 512 * i;
      const int tp = idx >> 10, q = (idx >> 8) & 3, rem = idx & 255;      // quadrant q = qi + 2 qo
      const int r = rem >> 4, l4 = (rem & 15) * 4;
      const int tap = p * 2 + tp;
      const int ci = ci0 + (q & 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l4 >> 5);
      const int co = co0 + (q >> 1) * 32 + (l4 & 31);
      if (tap < NT && ci < g.cin && co < g.cout) {
        const f32x4 sum = *reinterpret_cast<const f32x4*>(red + ((tp * 8 + q) * 16 + r) * 64 + l4) +
                          *reinterpret_cast<const f32x4*>(red + ((tp * 8 + q + 4) * 16 + r) * 64 + l4);
        float* dst = out + ((size_t)tap * g.cin + ci) * g.cout + co;
        if (g.vec_ok) {
          *reinterpret_cast<f32x4*>(dst) = sum;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (co + j < g.cout) dst[j] = sum[j];
        }
      }
    }
  }
  if constexpr (BIAS) {
    if (BIAS && ci_blk == 0) {      // workgroup-uniform; accb[0] on lanes 0..31 = column sum of co0 + 32 qo + lane
      __syncthreads();
      if (qi == 0 && lane < 32) red[wid * 32 + lane] = accb[0];
      __syncthreads();
      if (tid < 64 && co0 + tid < g.cout) {      // waves (qi 0, qo, kh): wid = 2 qo + 4 kh
        const int w0 = 2 * (tid >> 5);
        atomicAdd(g.gbias + co0 + tid, red[w0 * 32 + (tid & 31)] + red[(w0 + 4) * 32 + (tid & 31)]);
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------------------------
// Thin layers (cin or cout = 16: the 256x256 blocks of E / D, the generator's last concat conv).  The 32 x 32
// accumulator block of the kernel above is 75 % padding there (kbench: 16->16 and 16->32 at 256x256 take the same
// 100 us -- bound by MFMA issue and per-tile overhead, not by HBM).  Here the block is 16 ci x CO co (CO = 16 | 32)
// (one workgroup per 16 x 16 block of the weight) on v_mfma_f32_16x16x32_bf16: A = x^T [16 ci x 32 px], B = gy [32 px x 16 co], K = 32 pixels = the same two
// transpose reads per operand.  LDS holds 16 channels per pixel (32 B; gy: CO * 2 B), the tile is 16 rows x 16 cols
// (halo 18 x 18: 1.27x instead of 1.41x), wave w reduces row pairs (4w, 4w+1) and (4w+2, 4w+3) = 2 K steps per tile.
// K index k = (group gq = lane >> 4, e = 0..7)  <->  pixel (row r + (e >> 2), col 4*gq + (e & 3)): a half-wave then
// reads 8 consecutive pixels = 256 contiguous bytes per transpose read (conflict-free), same mapping for both operands.
// Accumulators: 9 x f32x4 instead of 9 x f32x16 -> 4 workgroups per CU instead of 2.  A 32-channel cout runs as two
// co blocks (a single workgroup with both, CO = 32, needs 214 VGPRs: 2 workgroups per CU, 112 vs 105 us in kbench).
template <int CO, bool BIAS, bool F16 = false>
__global__ __launch_bounds__(256, BIAS ? 3 : 4) void conv_wgrad_thin_kernel(const bf16* __restrict__ x, const bf16* __restrict__ gy,
                                                                 float* __restrict__ slab, const WgGeom g) {
  constexpr int TH = 16, TW = 16, HWX = 18, HH = 18, NT = 9, NB = CO / 16;
  constexpr int PSX = 32, PSG = CO * 2;
  constexpr int XPX = HH * HWX;                                         // 324 halo pixels
  constexpr int XVEC = XPX * 2, XSLOTS = (XVEC + 255) / 256;            // 648 16-byte vectors -> 3
  constexpr int GVEC = 256 * (CO / 8), GSLOTS = GVEC / 256;             // 2 | 4
  constexpr int X_BYTES = XPX * PSX;                                    // 10368
  constexpr int G_BYTES = 256 * PSG;                                    // 8192 | 16384
  typedef __attribute__((ext_vector_type(4))) float f32x4;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  int slice, pair;      // same XCD-aware (slice, ci block) mapping as conv_wgrad_tile_kernel
  {
    const int id = blockIdx.x, npairs = g.n_pairs;
    if ((g.nslices & 7) == 0) {
      const int hi = id / (8 * npairs), rem = id - hi * 8 * npairs;
      pair = rem >> 3;
      slice = hi * 8 + (rem & 7);
    } else {
      pair = id % npairs;
      slice = id / npairs;
    }
  }
  const int ci_blk = pair / g.n_co_blk, co_blk = pair - ci_blk * g.n_co_blk;
  const int ci0 = ci_blk * 16, co0 = co_blk * CO;

  int x_loff[XSLOTS], x_hy1[XSLOTS], x_hx1[XSLOTS], x_rel[XSLOTS];
  bool x_use[XSLOTS];
#pragma unroll
  for (int s = 0; s < XSLOTS; ++s) {
    const int v = tid + s * 256;
    const int px = v >> 1, part = v & 1;
    x_hy1[s] = px / HWX - 1;
    x_hx1[s] = px % HWX - 1;
    x_use[s] = v < XVEC;
    if (g.c0 == 0)
      x_rel[s] = ((x_hy1[s] * g.w + x_hx1[s]) * g.cin + ci0 + part * 8) * 2;
    else if (ci0 < g.c0)      // half-resolution source: (iy >> 1, ix >> 1); tile origins are even
      x_rel[s] = (((x_hy1[s] >> 1) * (g.w >> 1) + (x_hx1[s] >> 1)) * g.c0 + ci0 + part * 8) * 2;
    else                      // skip source
      x_rel[s] = ((x_hy1[s] * g.w + x_hx1[s]) * (g.cin - g.c0) + ci0 - g.c0 + part * 8) * 2;
    x_loff[s] = px * PSX + part * 16;
  }
  int g_loff[GSLOTS];
  unsigned g_rel[GSLOTS];
#pragma unroll
  for (int s = 0; s < GSLOTS; ++s) {
    const int v = tid + s * 256;
    const int px = v / (CO / 8), part = v % (CO / 8);
    const bool use = co0 + part * 8 + 8 <= g.cout;
    g_rel[s] = use ? (unsigned)((((px >> 4) * g.w + (px & 15)) * g.cout + co0 + part * 8) * 2) : WOOB;
    g_loff[s] = px * PSG + part * 16;
  }

  // fragment addresses: group gq points at pixel col 4*gq + (t16 >> 2) of the K step's first row; the second
  // transpose read is the same columns one row down
  const int gq = lane >> 4, t16 = lane & 15;
  const int fx = (gq * 4 + (t16 >> 2)) * PSX + (t16 & 3) * 8;
  const int fg = (gq * 4 + (t16 >> 2)) * PSG + (t16 & 3) * 8;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  auto frag = [&](const unsigned char* p, int row_stride) __attribute__((always_inline)) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + row_stride));
    s16x8 v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8, v);
  };

  f32x4 acc[NT][NB];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][b][j] = 0.f;
  f32x4 accb[NB];
  bf16x8 ones;
  if constexpr (BIAS) {
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int j = 0; j < 4; ++j) accb[b][j] = 0.f;
    {      // 1.0 in the element format, eight times
      typedef __attribute__((ext_vector_type(4))) unsigned u4;
      u4 o4;
      o4[0] = o4[1] = o4[2] = o4[3] = ones16x2<F16>();
      ones = __builtin_bit_cast(bf16x8, o4);
    }
  }
  const bool do_bias = BIAS && ci_blk == 0;      // uniform
  bool bias_a = false, bias_b = false, bias_0 = false, bias_1 = false;

  const bool from_up = g.c0 != 0 && ci0 < g.c0, from_skip = g.c0 != 0 && ci0 >= g.c0;
  const int xc = from_up ? g.c0 : (from_skip ? g.cin - g.c0 : g.cin);
  const size_t ximg = from_up ? (size_t)(g.h >> 1) * (g.w >> 1) * xc : (size_t)g.h * g.w * xc;
  const size_t gimg = (size_t)g.h * g.w * g.cout;
  const bf16* xsrc = from_skip ? g.x1 : x;
  const int tile_begin = slice * g.tiles_per_wg;
  int tile_end = tile_begin + g.tiles_per_wg;
  if (tile_end > g.total_tiles) tile_end = g.total_tiles;

  struct Stage {
    bf16x8 rx[XSLOTS], rg[GSLOTS];
  };
  // (per-tile divisions kept here: the tile cursor of conv_wgrad_tile_kernel made this HBM-bound kernel 5 % slower --
  // 111 vs 106 us at 16 -> 32 channels, 256 x 256, n = 64)
  auto load_tile = [&](Stage& st, int tile, bool& bias_on) __attribute__((always_inline)) {
    const unsigned live = tile < tile_end;
    int t = live ? tile : tile_begin;
    const bool segb = t >= g.tiles_a;
    bias_on = do_bias && ((g.bias_segs >> (segb ? 1 : 0)) & 1);
    if (segb) t -= g.tiles_a;
    const bf16* xs = segb ? g.xb : xsrc;
    const bf16* gs = segb ? g.gyb : gy;
    const int tx = t % g.tiles_x;
    t /= g.tiles_x;
    const int ty = t % g.tiles_y;
    const int img = t / g.tiles_y;
    const int ox0 = tx * TW, oy0 = ty * TH;
    const int ximg_i = (from_skip && g.gsz) ? (int)((g.perm >> (8 * (img / g.gsz))) & 0xffu) * g.gsz + img % g.gsz : img;
    const __amdgpu_buffer_rsrc_t bx = wg_rsrc(xs + (size_t)ximg_i * ximg, (unsigned)(ximg * 2));
    const __amdgpu_buffer_rsrc_t bg = wg_rsrc(gs + (size_t)img * gimg, (unsigned)(gimg * 2));
    const int xbase = from_up ? ((oy0 >> 1) * (g.w >> 1) + (ox0 >> 1)) * xc * 2 : (oy0 * g.w + ox0) * xc * 2;
    const unsigned gbase = (unsigned)((oy0 * g.w + ox0) * g.cout * 2);
#pragma unroll
    for (int s = 0; s < XSLOTS; ++s) {
      const unsigned ok = live & (unsigned)x_use[s] & (unsigned)((unsigned)(oy0 + x_hy1[s]) < (unsigned)g.h) &
                          (unsigned)((unsigned)(ox0 + x_hx1[s]) < (unsigned)g.w);
      const unsigned off = ok ? (unsigned)(xbase + x_rel[s]) : WOOB;
      st.rx[s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(bx, off, 0, 0));
    }
#pragma unroll
    for (int s = 0; s < GSLOTS; ++s)
      st.rg[s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(bg, live ? gbase + g_rel[s] : WOOB, 0, 0));
  };
  auto stage_to_lds = [&](const Stage& st, unsigned char* bX, unsigned char* bG) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < XSLOTS; ++s)
      if (s < XSLOTS - 1 || tid + s * 256 < XVEC) *reinterpret_cast<bf16x8*>(bX + x_loff[s]) = st.rx[s];
#pragma unroll
    for (int s = 0; s < GSLOTS; ++s) *reinterpret_cast<bf16x8*>(bG + g_loff[s]) = st.rg[s];
  };
  auto reduce_tile = [&](const unsigned char* bX, const unsigned char* bG, bool bias_on) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int r = wid * 4 + ks * 2;      // first output row of this K step
      bf16x8 gf[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) gf[b] = frag(bG + r * 16 * PSG + fg + b * 32, 16 * PSG);
      if constexpr (BIAS) {
        if (bias_on) {
#pragma unroll
          for (int b = 0; b < NB; ++b) accb[b] = mfma_16x16x32<F16>(ones, gf[b], accb[b]);
        }
      }
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const bf16x8 xf = frag(bX + ((r + ky) * HWX + kx) * PSX + fx, HWX * PSX);
#pragma unroll
          for (int b = 0; b < NB; ++b)
            acc[ky * 3 + kx][b] = mfma_16x16x32<F16>(xf, gf[b], acc[ky * 3 + kx][b]);
        }
      }
    }
  };

  unsigned char* sX = wg_smem;
  unsigned char* sG = wg_smem + X_BYTES;
  unsigned char* sX1 = wg_smem + X_BYTES + G_BYTES;
  unsigned char* sG1 = sX1 + X_BYTES;
  Stage sa, sb;
  load_tile(sa, tile_begin, bias_a);
  load_tile(sb, tile_begin + 1, bias_b);
  for (int tile = tile_begin; tile < tile_end; tile += 2) {
    stage_to_lds(sa, sX, sG);
    bias_0 = bias_a;
    __syncthreads();
    load_tile(sa, tile + 2, bias_a);
    reduce_tile(sX, sG, bias_0);
    stage_to_lds(sb, sX1, sG1);
    bias_1 = bias_b;
    __syncthreads();
    load_tile(sb, tile + 3, bias_b);
    reduce_tile(sX1, sG1, bias_1);
  }

  // ---- cross-wave reduction through LDS, one co half at a time (4 waves x 36 regs x 64 lanes floats = 36 KiB), and
  // the slab write.  acc[tap][b][r]: ci = ci0 + 4*(lane >> 4) + r, co = 16*b + (lane & 15)
  float* red = reinterpret_cast<float*>(wg_smem);
  constexpr int NR = NT * 4;                               // registers per lane and co half
  float* out = slab + (size_t)slice * NT * g.cin * g.cout;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < NT; ++tap)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wid * NR + tap * 4 + r) * 64 + lane] = acc[tap][b][r];
    __syncthreads();
    for (int i = tid; i < NR * 64; i += 256) {
      const int q = i >> 6, l2 = i & 63;
      const float sum = red[(0 * NR + q) * 64 + l2] + red[(1 * NR + q) * 64 + l2] + red[(2 * NR + q) * 64 + l2] +
                        red[(3 * NR + q) * 64 + l2];
      const int r = q & 3, tap = q >> 2;
      const int ci = ci0 + 4 * (l2 >> 4) + r, co = co0 + 16 * b + (l2 & 15);
      if (co < g.cout) out[((size_t)tap * g.cin + ci) * g.cout + co] = sum;
    }
  }
  if constexpr (BIAS) {
    if (do_bias) {      // accb[b][0] on lanes 0..15 = row 0 of the all-equal rows: column sum of co 16*b + lane
      __syncthreads();
      if (lane < 16) {
#pragma unroll
        for (int b = 0; b < NB; ++b) red[wid * 32 + b * 16 + lane] = accb[b][0];
      }
      __syncthreads();
      if (tid < CO && co0 + tid < g.cout) atomicAdd(g.gbias + co0 + tid, red[tid] + red[32 + tid] + red[64 + tid] + red[96 + tid]);
    }
  }
}

// gw[i] (+)= sum over the k-slices of slab[k][i].  A workgroup owns 256/SG consecutive elements; its 256
// threads are SG slice groups x 256/SG elements, summed through LDS: no pre-zeroing; one
// atomic per element only when accumulating into an existing gradient (several streams may feed one sink).  SG is large for small weights (few elements, many slices) and 1 for large ones.
template <int SG>
__global__ __launch_bounds__(256) void conv_wgrad_slab_reduce(const float* __restrict__ slab, float* __restrict__ gw,
                                                              int64_t nw, int nslices, int accumulate) {
  constexpr int EPB = 256 / SG;
  __shared__ float part[SG][EPB + 1];
  const int e = threadIdx.x % EPB, sg = threadIdx.x / EPB;
  const int64_t i = (int64_t)blockIdx.x * EPB + e;
  // 8 independent partial sums in a fixed order: 8 loads in flight per thread (with 2 the kernel ran at one L2 round
  // trip per pair of slices: 11.8 us for 1024 slabs of a 16 x 16 layer, rocprofv3) and a result that does not depend
  // on timing
  float a[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) a[u] = 0.f;
  if (i < nw) {
    int k = sg;
    for (; k + 7 * SG < nslices; k += 8 * SG) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += slab[(size_t)(k + u * SG) * nw + i];
    }
    for (; k < nslices; k += SG) a[0] += slab[(size_t)k * nw + i];
  }
  const float s0 = (a[0] + a[1]) + (a[2] + a[3]), s1 = (a[4] + a[5]) + (a[6] + a[7]);
  if (SG == 1) {
    if (i < nw) {
      if (accumulate) atomicAdd(gw + i, s0 + s1);      // sinks may be fed from several streams at once
      else gw[i] = s0 + s1;
    }
    return;
  }
  part[sg][e] = s0 + s1;
  __syncthreads();
  if (threadIdx.x < EPB && i < nw) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < SG; ++j) t += part[j][e];
    if (accumulate) atomicAdd(gw + i, t);              // sinks may be fed from several streams at once
    else gw[i] = t;
  }
}

bool wg_thin(int h, int w, int cin, int cout);

// Waves per workgroup of conv_wgrad_tile_kernel for this layer: 8 (16-row tiles, one workgroup per CU, half the slabs
// for the same waves in flight) or 4.  Measured on one box, n = 64, us with 4 / 8 waves (gpurun_out r3d): the encoder /
// discriminator layers (cin <= cout) gain where the slice count stays a multiple of 8 (the XCD co-scheduling of the
// blocks that share a 128-byte line needs that) or the map is one tile -- 128^2 32>32 39.8 / 34.9, 64^2 64>64 34.7 / 33.1,
// 32^2 128>128 33.3 / 31.9, 16^2 256>256 37.3 / 34.2, with the bias gradient 37.5 / 32.8 ... 35.0 / 29.9; widening ones are
// even (128^2 32>64 55 / 53, 64^2 64>128 52.0 / 52.2, 32^2 128>256 50.4 / 51.3).  The generator's narrowing layers LOSE:
// 32^2 512>128 89 / 111 (4 slices: no co-scheduling), 64^2 256>64 90 / 95, 128^2 128>32 101 / 112, 16^2 512>256 53.8 / 55.1.
int wg_waves(int h, int w, int cin, int cout) {
  if (w == 8 || h % 16 != 0 || w % 16 != 0 || wg_thin(h, w, cin, cout)) return 4;
  if (cin > cout) return 4;
  const int pairs = ((cin + 31) / 32) * ((cout + 31) / 32);
  const int slices8 = pairs <= 256 ? 256 / pairs : 1;
  return (slices8 % 8 == 0 || h <= 16) ? 8 : 4;
}

// The quadrant kernel takes 3x3 layers whose channel counts are multiples of 64 (whole 128-byte lines per pixel; a concat
// input must split on a 64-channel boundary).  Measured against the tile kernel on one box, n = 64, us tile / quadrant
// (gpurun_out r3g): its loop is ~8 % faster where the slabs do not matter -- 32^2 512>128 90.0 / 82.2 (940 TFLOP/s), 64^2
// 256>64 87.4 / 81.0, 16^2 512>256 53.9 / 52.2, 32^2 128>256 50.2 / 48.8 -- but a workgroup's slab is four times a tile
// workgroup's (147 KB), so the layers with few blocks and short loops lose: 64^2 64>64 33.0 / 36.0, 32^2 128>128 32.1 / 34.6,
// 16^2 256>256 34.1 / 39.3 (n = 16: 21.4 / 32.2), and with the fused bias gradient (254 VGPRs) 29.2 / 38.2.  Taken where it
// measured faster: narrowing layers (the generator's concat convs), and widening ones with >= 8 blocks and >= 16 tiles per
// workgroup; never with the bias gradient.
bool wg_quad(int h, int w, int cin, int cout, int c0, int total_tiles8, bool bias) {
  if (w % 16 != 0 || h % 8 != 0 || cin % 64 != 0 || cout % 64 != 0 || c0 % 64 != 0) return false;
  if (bias) return false;
  if (cin > cout) return true;
  const int pairs = (cin / 64) * (cout / 64);
  const int slices = pairs <= 256 ? 256 / pairs : 1;
  return cin < cout && pairs >= 8 && total_tiles8 / slices >= 16;
}

void wg_split(int n, int h, int w, int cin, int cout, WgGeom* g, int* nslices, int nb = 0, int c0 = 0, bool allow_quad = true,
              bool bias = false) {
  const bool thin = wg_thin(h, w, cin, cout);
  g->quad = (allow_quad && w != 8 && wg_quad(h, w, cin, cout, c0, (w / 16) * (h / 8) * (n + nb), bias)) ? 1 : 0;
  g->vec_ok = 0;
  g->nw = g->quad ? 8 : wg_waves(h, w, cin, cout);
  g->n = n; g->h = h; g->w = w; g->cin = cin; g->cout = cout;
  g->x1 = nullptr;
  g->c0 = g->gsz = 0;
  g->perm = 0;
  g->xb = g->gyb = nullptr;
  g->nb = nb;
  g->gbias = nullptr;
  g->bias_segs = 3;
  if (w == 8) {      // 8x8 maps: a tile is a pair of images
    g->tiles_x = g->tiles_y = 1;
    g->tiles_a = (n + 1) / 2;
    g->total_tiles = g->tiles_a + (nb + 1) / 2;
  } else {
    g->tiles_x = w / 16;
    g->tiles_y = h / ((thin || (g->nw == 8 && !g->quad)) ? 16 : 8);
    g->tiles_a = g->tiles_x * g->tiles_y * n;
    g->total_tiles = g->tiles_a + g->tiles_x * g->tiles_y * nb;
  }
  const int n_ci = g->quad ? cin / 64 : (thin ? cin / 16 : (cin + 31) / 32);      // thin: 16-channel ci blocks, one co block (cout <= 32)
  g->n_co_blk = g->quad ? cout / 64 : (thin ? (cout + 15) / 16 : (cout + 31) / 32);
  // ~2 workgroups per CU in total; 1 per CU when that leaves a workgroup fewer than 8 tiles: every workgroup
  // costs one slab write + read (9*32*32 floats), which then outweighs its share of the input traffic
  // (kbench, 128x128x32 n16: 24.2 -> 20.6 us; 256x256x16 n48 prefers 512: 70 vs 85 us).
  const int pairs = n_ci * g->n_co_blk;
  int want = (thin ? 1024 : (g->nw == 8 ? 256 : 512)) / pairs;      // thin: 4 workgroups per CU fit; 8 waves: one
  if (want < 1) want = 1;
  if (g->nw != 8 && (g->total_tiles + want - 1) / want < (thin ? 4 : 8)) want = 256 / pairs;
  if (want < 1) want = 1;
  if (want > g->total_tiles) want = g->total_tiles;
  g->tiles_per_wg = (g->total_tiles + want - 1) / want;
  *nslices = (g->total_tiles + g->tiles_per_wg - 1) / g->tiles_per_wg;
  g->n_pairs = n_ci * g->n_co_blk;
  g->nslices = *nslices;
}

}  // namespace

int tg_wgrad_slab_reduce(const float* slab, float* gw, int64_t nw, int nslices, int accumulate, hipStream_t s);

namespace {
bool wg_thin(int h, int w, int cin, int cout) {
  return h % 16 == 0 && w % 16 == 0 && h >= 64 && cin % 16 == 0 && cout % 8 == 0 && cout <= 32 && (cin == 16 || cout == 16);
}

// launches the thin kernel for geometry g (already split); false: not a thin layer
bool wg_launch_thin(const WgGeom& g, const bf16* x, const bf16* gy, float* ws, int nslices, hipStream_t s) {
  if (!wg_thin(g.h, g.w, g.cin, g.cout)) return false;
  const dim3 grid(nslices * (g.cin / 16) * g.n_co_blk);
  const bool bias = g.gbias != nullptr;
  const size_t lds = 2 * (18 * 18 * 32 + 256 * 32);
  tg_note_kernel(bias ? "conv_wgrad_thin_kernel<16,bias>" : "conv_wgrad_thin_kernel<16>");
  if (bias && tg_elem_f16()) hipLaunchKernelGGL((conv_wgrad_thin_kernel<16, true, true>), grid, dim3(256), lds, s, x, gy, ws, g);
  else if (bias) hipLaunchKernelGGL((conv_wgrad_thin_kernel<16, true>), grid, dim3(256), lds, s, x, gy, ws, g);
  else if (tg_elem_f16()) hipLaunchKernelGGL((conv_wgrad_thin_kernel<16, false, true>), grid, dim3(256), lds, s, x, gy, ws, g);
  else hipLaunchKernelGGL((conv_wgrad_thin_kernel<16, false>), grid, dim3(256), lds, s, x, gy, ws, g);
  return true;
}
// launches conv_wgrad_tile_kernel for geometry g (already split, pointers and bias fields set)
int wg_launch_tile(const WgGeom& g0, const bf16* x, const bf16* gy, float* ws, int nslices, hipStream_t s) {
  WgGeom g = g0;
  g.vec_ok = (g.cout % 4 == 0 && (reinterpret_cast<uintptr_t>(ws) & 15) == 0) ? 1 : 0;
  const int n_ci = (g.cin + 31) / 32;
  const dim3 grid(nslices * n_ci * g.n_co_blk);
  // two tile buffers; the first bytes double as the cross-wave reduction scratch (16 KiB for 4 waves, 32 KiB for 8)
  const size_t lds8 = 2 * (2 * 10 * 10 * 64 + 8 * 16 * 64);        // 8x8 maps: a pair of images per tile
  const size_t lds16 = 2 * (10 * 18 * 64 + 8 * 16 * 64);           // 8 x 16 tiles, 4 waves
  const size_t lds16w8 = 2 * (18 * 18 * 64 + 16 * 16 * 64);        // 16 x 16 tiles, 8 waves: 74 240 B
  const bool bias = g.gbias != nullptr;
  if (g.quad) {
    const size_t ldsq = 2 * (2 * 10 * 18 * 64 + 2 * 8 * 16 * 64);      // two buffers of two x planes + two gy planes: 78 848 B
    static unsigned long long raised_q = 0;      // one bit per device
    if (tg_first_on_device(&raised_q)) {
      const void* ks[4] = {reinterpret_cast<const void*>(conv_wgrad_quad_kernel<true, true>),
                           reinterpret_cast<const void*>(conv_wgrad_quad_kernel<true, false>),
                           reinterpret_cast<const void*>(conv_wgrad_quad_kernel<false, true>),
                           reinterpret_cast<const void*>(conv_wgrad_quad_kernel<false, false>)};
      for (const void* k : ks) {
        if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsq) != hipSuccess) {
          tg_set_error("conv_wgrad_quad: cannot raise dynamic LDS to %zu", ldsq);
          return TG_ELAUNCH;
        }
      }
    }
    const dim3 gridq(nslices * g.n_pairs);
    tg_note_kernel("conv_wgrad_quad_kernel");
    const bool f16 = tg_elem_f16();
    if (bias && f16) hipLaunchKernelGGL((conv_wgrad_quad_kernel<true, true>), gridq, dim3(512), ldsq, s, x, gy, ws, g);
    else if (bias) hipLaunchKernelGGL((conv_wgrad_quad_kernel<true, false>), gridq, dim3(512), ldsq, s, x, gy, ws, g);
    else if (f16) hipLaunchKernelGGL((conv_wgrad_quad_kernel<false, true>), gridq, dim3(512), ldsq, s, x, gy, ws, g);
    else hipLaunchKernelGGL((conv_wgrad_quad_kernel<false, false>), gridq, dim3(512), ldsq, s, x, gy, ws, g);
    TG_LAUNCH_CHECK("conv_wgrad_quad");
    return TG_OK;
  }
  if (g.w == 8) {
    tg_note_kernel("conv_wgrad_tile_kernel");
    if (bias) TG_WG_LAUNCH(8, true, 4, grid, dim3(256), lds8, s, x, gy, ws, g);
    else TG_WG_LAUNCH(8, false, 4, grid, dim3(256), lds8, s, x, gy, ws, g);
  } else if (g.nw == 8) {
    static unsigned long long raised = 0;      // > 64 KiB of dynamic LDS: raise the limit of the four instantiations once per device
    if (tg_first_on_device(&raised)) {
      const void* ks[4] = {reinterpret_cast<const void*>(conv_wgrad_tile_kernel<16, true, true, 8>),
                           reinterpret_cast<const void*>(conv_wgrad_tile_kernel<16, true, false, 8>),
                           reinterpret_cast<const void*>(conv_wgrad_tile_kernel<16, false, true, 8>),
                           reinterpret_cast<const void*>(conv_wgrad_tile_kernel<16, false, false, 8>)};
      for (const void* k : ks) {
        if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16w8) != hipSuccess) {
          tg_set_error("conv_wgrad_tile: cannot raise dynamic LDS to %zu", lds16w8);
          return TG_ELAUNCH;
        }
      }
    }
    tg_note_kernel("conv_wgrad_tile_kernel<8 waves>");
    if (bias) TG_WG_LAUNCH(16, true, 8, grid, dim3(512), lds16w8, s, x, gy, ws, g);
    else TG_WG_LAUNCH(16, false, 8, grid, dim3(512), lds16w8, s, x, gy, ws, g);
  } else {
    tg_note_kernel("conv_wgrad_tile_kernel");
    if (bias) TG_WG_LAUNCH(16, true, 4, grid, dim3(256), lds16, s, x, gy, ws, g);
    else TG_WG_LAUNCH(16, false, 4, grid, dim3(256), lds16, s, x, gy, ws, g);
  }
  TG_LAUNCH_CHECK("conv_wgrad_tile");
  return TG_OK;
}
}  // namespace

bool tg_wgrad_tile_supported(int h, int w, int hout, int wout, int kh, int kw, int pad_t, int pad_l) {
  return kh == 3 && kw == 3 && pad_t == 1 && pad_l == 1 && h == hout && w == wout &&
         (((h % 8 == 0) && (w % 16 == 0)) || (h == 8 && w == 8));
}

// enough for whichever kernel the run picks (a concat input that does not split on a 64-channel boundary falls back
// from the quadrant kernel to the tile kernel, with another slice count)
size_t tg_wgrad_tile_workspace(int n, int h, int w, int cin, int cout) {
  WgGeom g;
  int nslices, nslices_t;
  wg_split(n, h, w, cin, cout, &g, &nslices);
  wg_split(n, h, w, cin, cout, &g, &nslices_t, 0, 0, false);
  return (size_t)(nslices > nslices_t ? nslices : nslices_t) * 9 * cin * cout * sizeof(float);
}

int tg_wgrad_tile_run(int n, int h, int w, int cin, int cout, const void* x, const void* gy, float* gw, int accumulate,
                      void* ws, size_t ws_bytes, hipStream_t s, float* gbias) {
  WgGeom g;
  int nslices;
  wg_split(n, h, w, cin, cout, &g, &nslices, 0, 0, true, gbias != nullptr);
  g.gbias = gbias;
  const int64_t nw = (int64_t)9 * cin * cout;
  TG_CHECK(ws && ws_bytes >= (size_t)nslices * nw * sizeof(float), TG_EINVAL,
           "tg_conv2d_bwd_weight(tile): workspace too small (%zu < %zu)", ws_bytes, (size_t)nslices * nw * sizeof(float));
  if (wg_launch_thin(g, (const bf16*)x, (const bf16*)gy, (float*)ws, nslices, s)) {
    TG_LAUNCH_CHECK("conv_wgrad_thin");
    return tg_wgrad_slab_reduce((const float*)ws, gw, nw, nslices, accumulate, s);
  }
  int rc = wg_launch_tile(g, (const bf16*)x, (const bf16*)gy, (float*)ws, nslices, s);
  if (rc) return rc;
  return tg_wgrad_slab_reduce((const float*)ws, gw, nw, nslices, accumulate, s);
}

size_t tg_wgrad_tile_workspace2(int na, int nb, int h, int w, int cin, int cout) {
  WgGeom g;
  int nslices, nslices_t;
  wg_split(na, h, w, cin, cout, &g, &nslices, nb);
  wg_split(na, h, w, cin, cout, &g, &nslices_t, nb, 0, false);
  return (size_t)(nslices > nslices_t ? nslices : nslices_t) * 9 * cin * cout * sizeof(float);
}

// gw (+)= wgrad(xa, gya) + wgrad(xb, gyb): two batches of the same layer in one launch
int tg_wgrad_tile_run2(int na, int nb, int h, int w, int cin, int cout, const void* xa, const void* gya, const void* xb,
                       const void* gyb, float* gw, int accumulate, void* ws, size_t ws_bytes, hipStream_t s, float* gbias,
                       int bias_segs) {
  WgGeom g;
  int nslices;
  wg_split(na, h, w, cin, cout, &g, &nslices, nb, 0, true, gbias != nullptr);
  g.xb = (const bf16*)xb;
  g.gyb = (const bf16*)gyb;
  g.gbias = gbias;
  g.bias_segs = bias_segs;
  const int64_t nw = (int64_t)9 * cin * cout;
  TG_CHECK(ws && ws_bytes >= (size_t)nslices * nw * sizeof(float), TG_EINVAL,
           "tg_conv2d_bwd_weight2: workspace too small (%zu < %zu)", ws_bytes, (size_t)nslices * nw * sizeof(float));
  if (wg_launch_thin(g, (const bf16*)xa, (const bf16*)gya, (float*)ws, nslices, s)) {
    TG_LAUNCH_CHECK("conv_wgrad_thin(2)");
    return tg_wgrad_slab_reduce((const float*)ws, gw, nw, nslices, accumulate, s);
  }
  int rc = wg_launch_tile(g, (const bf16*)xa, (const bf16*)gya, (float*)ws, nslices, s);
  if (rc) return rc;
  return tg_wgrad_slab_reduce((const float*)ws, gw, nw, nslices, accumulate, s);
}

// gw = d/dw of conv3x3_same(concat(nearest_up2(x0), x1)) read from the two sources (see conv_tile.hip UPCAT)
int tg_wgrad_tile_upcat_run(int n, int h, int w, int c0, int c1, int cout, int gsz, unsigned perm, const void* x0,
                            const void* x1, const void* gy, float* gw, int accumulate, void* ws, size_t ws_bytes,
                            hipStream_t s) {
  WgGeom g;
  int nslices;
  const int cin = c0 + c1;
  wg_split(n, h, w, cin, cout, &g, &nslices, 0, c0);
  g.x1 = (const bf16*)x1;
  g.c0 = c0;
  g.gsz = gsz;
  g.perm = perm;
  const int64_t nw = (int64_t)9 * cin * cout;
  TG_CHECK(ws && ws_bytes >= (size_t)nslices * nw * sizeof(float), TG_EINVAL,
           "tg_conv2d_bwd_weight_upcat: workspace too small (%zu < %zu)", ws_bytes, (size_t)nslices * nw * sizeof(float));
  if (wg_launch_thin(g, (const bf16*)x0, (const bf16*)gy, (float*)ws, nslices, s)) {
    TG_LAUNCH_CHECK("conv_wgrad_thin(upcat)");
    return tg_wgrad_slab_reduce((const float*)ws, gw, nw, nslices, accumulate, s);
  }
  int rc = wg_launch_tile(g, (const bf16*)x0, (const bf16*)gy, (float*)ws, nslices, s);
  if (rc) return rc;
  return tg_wgrad_slab_reduce((const float*)ws, gw, nw, nslices, accumulate, s);
}

// ---- deferred, batched slab reductions -------------------------------------------------------------------------------
// Between tg_wgrad_defer(1) and tg_wgrad_defer_flush the reductions that ACCUMULATE into a caller buffer (gradient sinks:
// nobody reads them before the optimiser) are not launched but queued; the flush issues all of them as ONE launch whose
// job table travels by value in the kernel arguments (so a hipGraph capture records it like any other launch).  86
// reductions of ~5 us plus a kernel boundary each leave a bench step (config 3; 114 in config 4).  Each job keeps the
// slice-group shape and the summation order of its stand-alone kernel.  The caller keeps the workspaces (slabs) alive
// until the flush and flushes on a stream that is ordered after every queued launch.
namespace {

struct SlabJob {
  const float* slab;
  float* gw;
  int nw, nslices, blk0, sg;      // elements, slices, first block of the job in the batched grid, slice groups (1 / 4 / 16)
};
constexpr int MAXJ = 120;         // 120 x 32 B + 8 B < the 4 KB kernel-argument segment
struct SlabJobTable {
  SlabJob j[MAXJ];
  int n, blocks;
};
// Host side, ONE queue per process: tg_wgrad_defer / _flush come from the trainer's thread, the queued reductions from the
// autograd engine's device thread that runs its backward nodes -- so not thread_local; a mutex orders them.  One trainer
// (one device) per process defers at a time: the one-process-per-GPU layout of DESIGN.md section 6.
SlabJobTable g_defer;
int g_defer_on = 0;
std::mutex g_defer_mu;

// VEC = 4: a thread owns FOUR consecutive elements (one 16-byte load per slice, 8 slices in flight: 128 bytes per thread
// against 32 with one element per thread; nw = 9 * cin * cout is a multiple of 4 and every slab row starts 16-byte aligned);
// VEC = 1: one element per thread (the stand-alone kernels' shape).  Per element the slices of a job are added in the
// stand-alone kernel's order either way.
// NO ATOMICS: the jobs that add into one sink (a discriminator kernel gets up to three filter gradients per backward: the
// batched pass, the gradient penalty's second pass and the pass back through the penalty's forward graph) form a CHAIN --
// the head owns the blocks, walks the chain and adds the jobs' sums to the sink in queueing order with a plain
// read-modify-write (nothing else touches a sink while the flush runs: the trainer joined its streams before it).  With one
// atomic per element and job the ORDER of three float adds depended on block scheduling -- the fp32 path's last bit moved
// from run to run (test_fp32_path_is_bit_reproducible, round 4) -- and ~14 M atomics per flush cost ~50 us.
// SlabJob::sg: bits 0-7 the slice groups (1 / 4 / 16), bits 16-31 the index of the next job of the chain (0: none; entry 0
// is always a head).  The table holds the `n` heads first (blk0 ascending), the chained jobs behind them.
template <int VEC>
__global__ __launch_bounds__(256) void conv_wgrad_slab_reduce_multi(const SlabJobTable tab) {
  typedef typename std::conditional<VEC == 4, f32x4, float>::type vt;
  __shared__ vt part[16 * 17 > 4 * 65 ? 16 * 17 : 4 * 65];
  int j = 0;
  while (j + 1 < tab.n && (int)blockIdx.x >= tab.j[j + 1].blk0) ++j;      // block-uniform
  vt* __restrict__ gw = reinterpret_cast<vt*>(tab.j[j].gw);
  const int64_t nwv = tab.j[j].nw / VEC;
  const int sgn = tab.j[j].sg & 0xff, epb = 256 / sgn;      // the same for every job of a chain (a function of nw)
  const int e = threadIdx.x % epb, sg = threadIdx.x / epb;
  const int64_t i = (int64_t)((int)blockIdx.x - tab.j[j].blk0) * epb + e;
  const bool owner = (int)threadIdx.x < epb && i < nwv;
  vt total = vt{};
  if (owner) total = gw[i];
  for (int job = j;;) {      // block-uniform chain walk
    const vt* __restrict__ slab = reinterpret_cast<const vt*>(tab.j[job].slab);
    const int nslices = tab.j[job].nslices;
    vt a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = vt{};
    if (i < nwv) {
      int k = sg;
      for (; k + 7 * sgn < nslices; k += 8 * sgn) {
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] += slab[(size_t)(k + u * sgn) * nwv + i];
      }
      for (; k < nslices; k += sgn) a[0] += slab[(size_t)k * nwv + i];
    }
    vt t = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    if (sgn != 1) {
      part[sg * (epb + 1) + e] = t;
      __syncthreads();
      if (owner) {
        t = vt{};
        for (int q = 0; q < sgn; ++q) t += part[q * (epb + 1) + e];
      }
      __syncthreads();      // the next job of the chain reuses the scratch
    }
    total += t;
    job = (int)((unsigned)tab.j[job].sg >> 16);
    if (!job) break;
  }
  if (owner) gw[i] = total;
}

}  // namespace

extern "C" int tg_wgrad_defer(int on) {
  std::lock_guard<std::mutex> lock(g_defer_mu);
  const int prev = g_defer_on;
  g_defer_on = on ? 1 : 0;
  if (!on) g_defer.n = g_defer.blocks = 0;      // whatever was not flushed is dropped (an abandoned pass)
  return prev;
}

extern "C" int tg_wgrad_defer_flush(void* stream) {
  std::lock_guard<std::mutex> lock(g_defer_mu);
  const int n = g_defer.n;
  if (n > 0) {
    // heads (the first job of every sink, in queueing order) in front, the other jobs of their chains behind them; grid laid
    // out over the heads for the vector width of this flush
    constexpr int vec = 4;
    SlabJobTable tab;
    int order[MAXJ], head_of[MAXJ], nheads = 0;
    for (int a = 0; a < n; ++a) {
      head_of[a] = a;
      for (int b = 0; b < a; ++b)
        if (g_defer.j[b].gw == g_defer.j[a].gw) {
          head_of[a] = head_of[b];
          break;
        }
      if (head_of[a] == a) order[nheads++] = a;
    }
    int pos[MAXJ], m = nheads;
    for (int h = 0; h < nheads; ++h) pos[order[h]] = h;
    for (int a = 0; a < n; ++a)
      if (head_of[a] != a) pos[a] = m++;
    int blocks = 0;
    for (int a = 0; a < n; ++a) {
      SlabJob jb = g_defer.j[a];
      jb.sg &= 0xff;
      int nxt = 0;      // the next job queued for the same sink
      for (int b = a + 1; b < n && !nxt; ++b)
        if (head_of[b] == head_of[a]) nxt = pos[b];
      jb.sg |= nxt << 16;
      jb.blk0 = 0;
      tab.j[pos[a]] = jb;
    }
    for (int h = 0; h < nheads; ++h) {
      SlabJob& jb = tab.j[h];
      jb.blk0 = blocks;
      const int epb = 256 / (jb.sg & 0xff);      // threads per slice group; each owns `vec` elements
      blocks += (int)((jb.nw / vec + epb - 1) / epb);
    }
    tab.n = nheads;
    tab.blocks = blocks;
    if (vec == 4)
      hipLaunchKernelGGL(conv_wgrad_slab_reduce_multi<4>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, tab);
    else
      hipLaunchKernelGGL(conv_wgrad_slab_reduce_multi<1>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, tab);
    g_defer.n = g_defer.blocks = 0;
    TG_LAUNCH_CHECK("conv_wgrad_slab_reduce_multi");
  }
  return n;
}

int tg_wgrad_slab_reduce(const float* slab, float* gw, int64_t nw, int nslices, int accumulate, hipStream_t s) {
  std::unique_lock<std::mutex> lock(g_defer_mu);
  if (g_defer_on && accumulate && g_defer.n < MAXJ && nw < (1ll << 31) && (nw & 3) == 0 &&
      (reinterpret_cast<uintptr_t>(slab) & 15u) == 0 && (reinterpret_cast<uintptr_t>(gw) & 15u) == 0 && !tg_deterministic_mode()) {
    SlabJob& jb = g_defer.j[g_defer.n++];
    jb.slab = slab;
    jb.gw = gw;
    jb.nw = (int)nw;
    jb.nslices = nslices;
    jb.sg = nw < 16384 ? 16 : (nw < 131072 ? 4 : 1);      // the stand-alone kernels' rule
    jb.blk0 = 0;                                           // the grid is laid out at the flush
    return TG_OK;
  }
  lock.unlock();
  if (nw < 16384)
    hipLaunchKernelGGL(conv_wgrad_slab_reduce<16>, dim3((unsigned)((nw + 15) / 16)), dim3(256), 0, s, slab, gw, nw, nslices,
                       accumulate);
  else if (nw < 131072)
    hipLaunchKernelGGL(conv_wgrad_slab_reduce<4>, dim3((unsigned)((nw + 63) / 64)), dim3(256), 0, s, slab, gw, nw, nslices,
                       accumulate);
  else
    hipLaunchKernelGGL(conv_wgrad_slab_reduce<1>, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, s, slab, gw, nw,
                       nslices, accumulate);
  TG_LAUNCH_CHECK("conv_wgrad_slab_reduce");
  return TG_OK;
}
