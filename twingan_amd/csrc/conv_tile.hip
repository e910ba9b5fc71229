// conv_tile: implicit-GEMM stride-1 SAME convolution (3x3 or 1x1) for feature maps of 16x16 and
// up, bf16 NHWC, fp32 accumulate on v_mfma_f32_32x32x16_bf16.  Forward and backward-data (the
// latter = this kernel over gy with the 180-degree rotated, transposed weight pack).
//
// Why a second kernel next to conv_mfma.hip: the first MFMA kernel spends ~1500 of its ~1650
// instructions per 128-pixel tile on runtime div/mod address arithmetic and is VALU-bound at
// 2 TB/s on the thin 128x128 / 256x256 layers.  Here every tile dimension is a compile-time
// constant, so halo / tap offsets fold into instruction immediates:
//
//   workgroup = 4 waves, output tile = (8*MT) rows x 16 cols of ONE image (MT = 1 or 2);
//   wave w owns MT sub-tiles of 2 rows x 16 cols = 32 pixels = the N dimension of one MFMA;
//   A operand = weights (32 output channels = M), B operand = pixels, so a lane's accumulator
//   holds 4 consecutive output channels of its pixel per register quad; lanes l and l+32 swap
//   quads (v_permlane32_swap) to own 16 consecutive channels -> 16-byte NHWC stores;
//   input halo tile [(8*MT+2)][18][KC] and weight tile [BN][taps*KC] live in LDS, both with one
//   16-byte pad per row so 16-byte fragment reads spread over all banks;
//   the next K chunk's global loads are issued before the current chunk's MFMAs (register
//   prefetch), written to LDS after them.
//
// Reference call sites replaced: tf.contrib.layers.conv2d at nets/pggan_utils.py:316-320 (every
// E/G/D conv of nets/pggan.py at 16x16 and above) and its Conv2DBackpropInput gradient.
#include "tg_common.h"
#include <cstdlib>

namespace {

struct TileGeom {
  int n, h, w, cin, cout;      // SAME conv: input and output are both [n,h,w,*]
  int cin_pad;                 // channels per tap in the weight pack (multiple of 16)
  int pad;                     // low-side zero padding (forward: (k-1)/2; backward-data: k-1-(k-1)/2)
  int tiles_x, tiles_y;        // tiles per image row / column
  int nblk;                    // tiles_x * tiles_y * n
  int tiles_per_wg;            // weight-resident variant: consecutive tiles per workgroup
  int epilogue;
  float alpha;
  // fused input concat(nearest_up2(x), x1) (UPCAT kernels): x is [n, h/2, w/2, c0], x1 is [n1, h, w, cin - c0];
  // output image i reads skip image perm-group(i) (see tg_upsample2x_concat_fwd)
  const bf16* x1;
  int c0, gsz;
  unsigned perm;
  // backward-data with the LeakyReLU backward of the PRODUCER of this conv's input folded into the epilogue:
  // out = acc * (mask > 0 ? 1 : alpha), mask = the forward input (= the producer's activation output), same shape as
  // the output (NULL: plain)
  const bf16* mask;
  // STATS kernels: per-workgroup sums of the (bf16-rounded) outputs and of their squares, per output channel, written
  // to stats[img][chunk][2][cout] -- what in_stats_partial<PART> (norm.hip) would have read the tensor back for.
  float* stats;
  int stat_chunks;         // chunks per image
  bool f16;                // 16-bit lanes are IEEE half (TG_F16) instead of bfloat16
  // POOL kernels: also write avg_pool2x2 of the (bf16-rounded) output, [n, h/2, w/2, cout] (the tf.nn.avg_pool that
  // ends a discriminator block, nets/pggan.py:304-306) -- the tile already holds every 2x2 block it needs
  bf16* ypool;
  // POOL kernels, optionally: instead of y itself write only the SIGN of each output, one bit per element
  // (ymask[n][h][w][cout/8] bytes, bit j of byte q = (y[.., 8q+j] > 0)) -- what the LeakyReLU backward of a
  // discriminator block's last conv needs of its full-resolution output when that output's only consumer is the pool
  // (nets/pggan.py:304-306): the tensor is never written (1/16 of its bytes instead) and never read back
  unsigned char* ymask;
  int* chunks_query;       // non-NULL: do not launch, report the chunk count the STATS variant of this dispatch would use
  // UPBWD kernels (MODE 3): backward-data of a conv whose input was concat(nearest_up2(x0), x1) (UPCAT forward), with the
  // adjoint of the upsample + concat -- tg_upsample2x_concat_bwd: a 2x2 SUM of the first c0 output channels, and the sum
  // over the generator groups that read one skip image of the others -- done in the epilogue, from the fp32 accumulators:
  // up_out [n, h/2, w/2, c0], skip_out [n1, h, w, cout - c0].  The concat-layout gradient tensor (537 MB at 256 x 256 x 64,
  // n 64) is never written.  Output-channel blocks below c0 ("up") run once per image; blocks from c0 on ("skip") run once
  // per SKIP image and walk the images of the groups og with perm[og] == that image's group (gsz, perm as in UPCAT).
  bf16* up_out;
  bf16* skip_out;
  int n1;
  // UNPOOL kernels: the conv input is never in memory -- it is the gradient of a discriminator block's last LeakyReLU layer,
  // x[n,y,x,c] = rnd(0.25 * up_src[n,y/2,x/2,c] * (sign bit of (n,y,x,c) ? 1 : up_alpha)), i.e. AvgPoolGrad + LeakyReluGrad
  // (tg_lrelu_pool_bwd_signs) applied while the halo tile is staged: up_src [n,h/2,w/2,cin] is the gradient of the pooled
  // output, up_signs [n,h,w,cin/8] the sign bytes tg_conv2d_fwd_pool_signs left.  Backward-data of that layer then reads
  // 1/4 + 1/16 of the tensor's bytes and the tensor itself is neither written nor read (nets/pggan.py:304-306).
  const bf16* up_src;
  const unsigned char* up_signs;
  const bf16* up_z;      // UNPOOL-Z kernels: the layer's activation output itself [n,h,w,cin] instead of its sign bytes (passes that
                         // kept it: the gradient penalty's, nets/pggan.py:304-306 under image_generation.py:414-439)
  float up_alpha;
  // ... and, when the layer's filter gradient needs that gradient tensor after all (a discriminator step), the kernel also
  // WRITES it: up_store [n,h,w,cin] receives every staged vector that is an interior pixel of its tile, from the workgroups
  // of output-channel block 0 (tiles partition the image, so each element is written exactly once) -- the separate
  // tg_lrelu_pool_bwd_signs launch and this kernel's read of its output are gone.  NULL: not written.
  bf16* up_store;
  // Weight-set groups (TgConvDesc::groups): npg > 0 = images per group -- image i is convolved with weight set i / npg, whose
  // pack starts wgs elements after the previous one (bias row: cout floats).  The two discriminators' layer as ONE launch
  // over [D_s rows; D_t rows].  conv_tile_kernel only: a workgroup works on one image.
  int npg;
  unsigned wgs;
};

// LDS layout of a staged halo tile: PS bytes per pixel, ROW bytes per halo row, such that the 16-byte fragment reads of a
// wave hit every bank once.  ds_read_b128 serves a wave in four groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19,
// 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS) -- and a group is conflict-free when its 16 x 4 dwords cover the 64
// banks.  With the 32x32x16 operand layout (lanes 0-15 / 16-31 = 16 pixels of two consecutive halo rows, lanes 32-63 the
// next 16 bytes) the plain "one 16-byte pad per pixel" layout (KC*2 + 16 bytes, rows back to back) is 2-way conflicted on
// every pixel-fragment read: SQ_LDS_BANK_CONFLICT was 32-39 % of SQ_LDS_IDX_ACTIVE in the tile kernels and 50 % in the
// thin-output ones (profiles/r05_m_lds_bank_conflicts.txt).  Conflict-free (searched over strides, checked for every tap
// offset): 32-channel chunks keep 80 bytes per pixel and start every halo row on a 256-byte boundary; 16-channel chunks
// drop the per-pixel pad (32 bytes) and pad each row by 16.
template <int KC, int HWX>
struct HaloLds {
  static constexpr int PS = KC == 16 ? 32 : KC * 2 + 16;
  static constexpr int ROW = KC == 16 ? HWX * 32 + 16 : ((HWX * (KC * 2 + 16) + 255) & ~255);
};
// ... and with the 16x16x32 operand layout of the thin-output kernels (lanes 0-15 = 16 pixels of ONE row, lane group q =
// lane / 16 its 16-byte quarter / its tap): 96 bytes per pixel, rows back to back.
constexpr int THIN_PS = 96;

extern __shared__ __attribute__((aligned(16))) unsigned char tile_smem[];

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

// Buffer resources: out-of-range offsets read as zero / drop the store, so image borders, unused
// staging slots and partial channel blocks need no exec-mask branches (offset OOB = "masked off").
constexpr unsigned OOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ bf16x8 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}

// LeakyReLU derivative of 4 consecutive bf16 activations (8 bytes at `off`): z > 0 ? 1 : alpha
__device__ __forceinline__ void mask4(__amdgpu_buffer_rsrc_t r, unsigned off, float alpha, float (&f)[4]) {
  const u32x2 z = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
  f[0] = (short)(z[0] & 0xffffu) > 0 ? 1.f : alpha;      // a positive bf16 is a positive int16 pattern
  f[1] = (short)(z[0] >> 16) > 0 ? 1.f : alpha;
  f[2] = (short)(z[1] & 0xffffu) > 0 ? 1.f : alpha;
  f[3] = (short)(z[1] >> 16) > 0 ? 1.f : alpha;
}


// UNPOOL staging: eight pooled-gradient values q and their eight sign bits -> rnd(0.25 * q * (bit ? 1 : alpha)), the
// arithmetic (and rounding) of lrelu_bwd_bias_kernel<.., BITS> in norm.hip, so the staged tile is bit-identical to the tensor
// tg_lrelu_pool_bwd_signs would have written
__device__ __forceinline__ unsigned buf_load_u8(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return (unsigned)__builtin_amdgcn_raw_buffer_load_b8(r, off, 0, 0);
}
// bit j = (element j > 0) of eight 16-bit activations (a positive bf16 / f16 is a positive int16 pattern)
__device__ __forceinline__ unsigned sign_bits8(bf16x8 z) {
  const u32x4 u = __builtin_bit_cast(u32x4, z);
  unsigned m = 0;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    m |= ((short)(u[d] & 0xffffu) > 0 ? 1u : 0u) << (2 * d);
    m |= ((short)(u[d] >> 16) > 0 ? 1u : 0u) << (2 * d + 1);
  }
  return m;
}
// (0.25 * q) * slope == q * (0.25 * slope): a scaling by a power of two commutes with the rounding of the product (the one
// difference: q = -0 stays -0 here and became +0 there -- equal as numbers, and as MFMA operands)
template <bool F16>
__device__ __forceinline__ bf16x8 unpool8(bf16x8 q, unsigned bits, float alpha) {
  const u32x4 u = __builtin_bit_cast(u32x4, q);
  const float s1 = 0.25f, sa = 0.25f * alpha;
  u32x4 o;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const float lo = unpack16_lo<F16>(u[d]) * (((bits >> (2 * d)) & 1u) ? s1 : sa);
    const float hi = unpack16_hi<F16>(u[d]) * (((bits >> (2 * d + 1)) & 1u) ? s1 : sa);
    o[d] = pack16x2<F16>(lo, hi);
  }
  return __builtin_bit_cast(bf16x8, o);
}

// bit j = (element j > 0) of the 16 packed 16-bit values in (a, b) (a positive bf16 / f16 is a positive int16 pattern)
__device__ __forceinline__ unsigned sign_bits16(u32x4 a, u32x4 b) {
  unsigned m = 0;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    m |= ((short)(a[d] & 0xffffu) > 0 ? 1u : 0u) << (2 * d);
    m |= ((short)(a[d] >> 16) > 0 ? 1u : 0u) << (2 * d + 1);
    m |= ((short)(b[d] & 0xffffu) > 0 ? 1u : 0u) << (8 + 2 * d);
    m |= ((short)(b[d] >> 16) > 0 ? 1u : 0u) << (8 + 2 * d + 1);
  }
  return m;
}

// Transposing sum over the 32 lanes of a half-wave, steps [S0, S1) of 5: entering step s a lane holds 32 >> s values;
// lanes whose bit s differs exchange halves, so after all five steps lane l31 holds the total of original value
// v[l31] in v[0].  31 exchanges for 32 values (a plain butterfly per value would take 160), fixed summation order.
template <int CTRL>
__device__ __forceinline__ float dpp_quad(float x) {      // quad_perm lane exchange: no LDS, folds into the consumer
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
template <int S>
__device__ __forceinline__ float lane_xor(float x) {
  if constexpr (S == 0) return dpp_quad<0xB1>(x);      // quad_perm [1,0,3,2]
  else if constexpr (S == 1) return dpp_quad<0x4E>(x); // quad_perm [2,3,0,1]
  else return __shfl_xor(x, 1 << S);
}
template <int S>
__device__ __forceinline__ void transpose_sum_step(float* v, int l31) {
  const bool hi = (l31 >> S) & 1;
#pragma unroll
  for (int i = 0; i < (32 >> (S + 1)); ++i) {
    const float a = v[2 * i], b = v[2 * i + 1];
    v[i] = (hi ? b : a) + lane_xor<S>(hi ? a : b);
  }
}
template <int S0, int S1>
__device__ __forceinline__ void half_wave_transpose_sum(float* v, int l31) {
  if constexpr (S0 <= 0 && 0 < S1) transpose_sum_step<0>(v, l31);
  if constexpr (S0 <= 1 && 1 < S1) transpose_sum_step<1>(v, l31);
  if constexpr (S0 <= 2 && 2 < S1) transpose_sum_step<2>(v, l31);
  if constexpr (S0 <= 3 && 3 < S1) transpose_sum_step<3>(v, l31);
  if constexpr (S0 <= 4 && 4 < S1) transpose_sum_step<4>(v, l31);
}
// steps 0 and 1 on four values: the lane's share of the quad total (lane bits 0,1 pick which of the four)
__device__ __forceinline__ float quad_fold4(float a0, float a1, float a2, float a3, int l31) {
  const bool h0 = l31 & 1, h1 = (l31 >> 1) & 1;
  const float b0 = (h0 ? a1 : a0) + dpp_quad<0xB1>(h0 ? a0 : a1);
  const float b1 = (h0 ? a3 : a2) + dpp_quad<0xB1>(h0 ? a2 : a3);
  return (h1 ? b1 : b0) + dpp_quad<0x4E>(h1 ? b0 : b1);
}


// 2x2 average over this lane's pixel quad -- lanes l31 ^ 1 (the neighbouring column) and l31 ^ 16 (the sub-tile's other
// row) -- of the 8 bf16-rounded values in p (four channel quads of one 32-channel block, as packed for the store); all
// four lanes of a quad receive the result.  Column pairs by a DPP quad exchange, row pairs by v_permlane16_swap.
__device__ __forceinline__ float add_lane_xor16(float x) {
  const unsigned u = __builtin_bit_cast(unsigned, x);
  auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);      // (rows 0,0,2,2 | rows 1,1,3,3) of x
  return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
template <bool F16>
__device__ __forceinline__ void pool_quad(const unsigned (&p)[4][2], unsigned (&pp)[4][2]) {
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      float lo = unpack16_lo<F16>(p[q][d]), hi = unpack16_hi<F16>(p[q][d]);
      lo += dpp_quad<0xB1>(lo);
      hi += dpp_quad<0xB1>(hi);
      pp[q][d] = pack16x2<F16>(0.25f * add_lane_xor16(lo), 0.25f * add_lane_xor16(hi));
    }
}

// Workgroup tail of the STATS kernels: every wave's lane l31 holds (for each 32-channel block nt) the half-wave total
// of statistic r = l31 -- r < 16: sum of channel 8*(r/4) + 4*kgrp + r%4, r >= 16: sum of squares of channel r - 16 --
// the four waves (different pixels, same channels) are added in wave order through LDS and written out.
template <int BN>
__device__ __forceinline__ void stats_flush(const float (&tot)[BN / 32], float* red, int tid, int n0, int cout,
                                            float* __restrict__ out) {
  const int lane = tid & 63, wid = tid >> 6, kgrp = lane >> 5, l31 = lane & 31;
  const int which = l31 >> 4, rr = l31 & 15;
  __syncthreads();                      // the staging buffers are free now
#pragma unroll
  for (int nt = 0; nt < BN / 32; ++nt) red[(wid * 2 + which) * BN + nt * 32 + (rr >> 2) * 8 + kgrp * 4 + (rr & 3)] = tot[nt];
  __syncthreads();
  if (tid < 2 * BN) {
    const int w2 = tid / BN, ch = tid % BN;
    const float t = (red[(0 * 2 + w2) * BN + ch] + red[(1 * 2 + w2) * BN + ch]) +
                    (red[(2 * 2 + w2) * BN + ch] + red[(3 * 2 + w2) * BN + ch]);
    if (n0 + ch < cout) out[(size_t)w2 * cout + n0 + ch] = t;
  }
}

// UPBWD: the gy images whose concat-layout gradient adds up into skip image `img1` -- packed as group indices, 8 bits
// each (<= 4 groups); without a group permutation the image itself.  -> number of sources (0: nobody read that image)
__device__ __forceinline__ int upbwd_sources(const TileGeom& g, int img1, unsigned* pk) {
  if (!g.gsz) {
    *pk = 0;
    return 1;
  }
  const int sg = img1 / g.gsz;
  int ns = 0;
  unsigned v = 0;
  for (int og = 0; og < g.n / g.gsz; ++og)
    if ((int)((g.perm >> (8 * og)) & 0xffu) == sg) v |= (unsigned)og << (8 * ns++);
  *pk = v;
  return ns;
}
__device__ __forceinline__ int upbwd_source_image(const TileGeom& g, int img1, unsigned pk, int s) {
  return g.gsz ? (int)((pk >> (8 * s)) & 0xffu) * g.gsz + img1 % g.gsz : img1;
}
// 2x2 sum over this lane's pixel quad (lanes l31 ^ 1 and l31 ^ 16, see pool_quad) of an fp32 value; every lane of the quad
// receives it
__device__ __forceinline__ float sum_quad(float v) {
  v += dpp_quad<0xB1>(v);
  return add_lane_xor16(v);
}

// UPCAT: the conv input is concat(nearest_up2(x), x1) on channels (generator_three_layer_block,
// nets/pggan.py:69-76) read straight from the two sources -- K chunks below c0 come from the half-resolution
// tensor, the rest from the skip tensor -- instead of from a materialised copy.
// MODE 0: plain; 1: statistics partials of the output (STATS); 2: also the 2x2 average pool of the output (POOL)
template <int KH, int KC, int BN, int MT, bool UPCAT = false, int MODE = 0, bool F16 = false>
__global__ __launch_bounds__(256) void conv_tile_kernel(const bf16* __restrict__ x, const bf16* __restrict__ wp0,
                                                        const float* __restrict__ bias0, bf16* __restrict__ y,
                                                        const TileGeom g) {
  constexpr bool STATS = MODE == 1, POOL = MODE == 2, UPBWD = MODE == 3, UNPOOL = MODE == 4 || MODE == 5, UPZ = MODE == 5;
  static_assert(!(UPBWD && UPCAT), "UPBWD is a backward-data mode: its input is the plain output gradient");
  static_assert(!(UNPOOL && UPCAT), "UNPOOL is a backward-data mode: its input is the pooled gradient + sign bytes");
  constexpr int KW = KH, NT = KH * KW;
  constexpr int TW = 16, TH = 8 * MT;
  constexpr int HWX = TW + KW - 1, HH = TH + KH - 1;
  constexpr int VPP = KC / 8;                       // 16-byte vectors per pixel per chunk
  constexpr int PS_A = HaloLds<KC, HWX>::PS;        // LDS bytes per halo pixel
  constexpr int ROW_A = HaloLds<KC, HWX>::ROW;      // ... per halo row (bank-conflict-free fragment reads: see HaloLds)
  constexpr int RS_B = NT * KC * 2 + 16;            // LDS bytes per weight row
  constexpr int AVEC = HH * HWX * VPP;              // halo vectors per chunk
  constexpr int ASLOTS = (AVEC + 255) / 256;
  constexpr int BVEC = BN * NT * VPP;
  constexpr int BSLOTS = (BVEC + 255) / 256;
  constexpr int NTILE = BN / 32;
  constexpr int A_BYTES = (HH * ROW_A + 15) & ~15;

  unsigned char* sA = tile_smem;
  unsigned char* sB = tile_smem + A_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;

  // ---- XCD-aware tile order: workgroup b runs on XCD b % 8; give each XCD a contiguous tile range
  int t = blockIdx.x;
  if ((g.nblk & 7) == 0) t = (t & 7) * (g.nblk >> 3) + (t >> 3);
  const int tile_in_img = t % (g.tiles_x * g.tiles_y);
  const int tx = t % g.tiles_x;
  t /= g.tiles_x;
  const int ty = t % g.tiles_y;
  const int img = t / g.tiles_y;
  const int ox0 = tx * TW, oy0 = ty * TH;
  const int n0 = blockIdx.y * BN;
  const int c1 = g.cin - g.c0;
  // weight set of this image (uniform: a workgroup works on one image); npg == 0: one set
  const int wset = g.npg ? (img >= g.npg) + (img >= 2 * g.npg) + (img >= 3 * g.npg) : 0;
  const bf16* __restrict__ wp = wp0 + (size_t)wset * g.wgs;
  const float* __restrict__ bias = bias0 + wset * g.cout;      // only read under TG_EPI_BIAS (a zero-sized resource otherwise)
  const size_t img_elems = UPCAT ? (size_t)(g.h / 2) * (g.w / 2) * g.c0
                                 : UNPOOL ? (size_t)(g.h / 2) * (g.w / 2) * g.cin : (size_t)g.h * g.w * g.cin;
  // UPBWD: a "skip" block (output channels >= c0) belongs to skip image `img` and reads nsrc gy images
  const bool skip_blk = UPBWD && n0 >= g.c0;
  unsigned srcpk = 0;
  int nsrc = 1;
  if constexpr (UPBWD) {
    if (skip_blk) {
      if (img >= g.n1) return;      // block-uniform, before any barrier
      nsrc = upbwd_sources(g, img, &srcpk);
    }
  }
  const __amdgpu_buffer_rsrc_t rx = make_rsrc((UNPOOL ? g.up_src : x) + (size_t)img * img_elems, (unsigned)(img_elems * 2));
  // UNPOOL: one sign byte per 8 channels; UPZ: the activation tensor itself (16 bytes per 8 channels)
  const size_t sign_bytes = UPZ ? (size_t)g.h * g.w * g.cin * 2 : (size_t)g.h * g.w * (g.cin >> 3);
  const __amdgpu_buffer_rsrc_t rsg =
      make_rsrc(UPZ ? (const unsigned char*)g.up_z + (size_t)img * sign_bytes
                    : UNPOOL ? g.up_signs + (size_t)img * sign_bytes : (const unsigned char*)x,
                UNPOOL ? (unsigned)sign_bytes : 0u);
  const size_t img1_elems = (size_t)g.h * g.w * c1;
  const int img1 = (UPCAT && g.gsz) ? (int)((g.perm >> (8 * (img / g.gsz))) & 0xffu) * g.gsz + img % g.gsz : img;
  const __amdgpu_buffer_rsrc_t rx1 =
      make_rsrc(UPCAT ? g.x1 + (size_t)img1 * img1_elems : x, UPCAT ? (unsigned)(img1_elems * 2) : 0u);
  const int wrow = NT * g.cin_pad;
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(wp + (size_t)n0 * wrow, (unsigned)((size_t)BN * wrow * 2));

  // ---- per-thread staging slots (compile-time trip counts, constant divisors); byte offsets
  unsigned a_goff[ASLOTS];     // inside the image at chunk 0, or OOB (zero fill: border / unused slot)
  unsigned a_goff1[(UPCAT || UNPOOL) ? ASLOTS : 1];      // UPCAT: the same pixel in the skip tensor; UNPOOL: its sign byte
  unsigned a_soff[UNPOOL ? ASLOTS : 1];                  // UNPOOL: where the staged vector goes in up_store (OOB: nowhere)
  int a_loff[ASLOTS];
#pragma unroll
  for (int s = 0; s < ASLOTS; ++s) {
    const int v = tid + s * 256;
    const int px = v / VPP, part = v % VPP;
    const int hy = px / HWX, hx = px % HWX;
    const int iy = oy0 + hy - g.pad, ix = ox0 + hx - g.pad;
    a_loff[s] = hy * ROW_A + hx * PS_A + part * 16;
    const bool ok = (v < AVEC) && iy >= 0 && iy < g.h && ix >= 0 && ix < g.w;
    if constexpr (UPCAT) {
      a_goff[s] = ok ? (unsigned)((((iy >> 1) * (g.w >> 1) + (ix >> 1)) * g.c0 + part * 8) * 2) : OOB;
      a_goff1[s] = ok ? (unsigned)(((iy * g.w + ix) * c1 + part * 8) * 2) : OOB;
    } else if constexpr (UNPOOL) {
      a_goff[s] = ok ? (unsigned)((((iy >> 1) * (g.w >> 1) + (ix >> 1)) * g.cin + part * 8) * 2) : OOB;
      a_goff1[s] = !ok ? OOB : UPZ ? (unsigned)(((iy * g.w + ix) * g.cin + part * 8) * 2) : (unsigned)((iy * g.w + ix) * (g.cin >> 3) + part);
      // interior pixels of the tile (the halo ring belongs to the neighbours): where the staged vector is also written
      const bool own = ok && hy >= g.pad && hy < g.pad + TH && hx >= g.pad && hx < g.pad + TW;
      a_soff[s] = own ? (unsigned)(((iy * g.w + ix) * g.cin + part * 8) * 2) : OOB;
    } else {
      a_goff[s] = ok ? (unsigned)(((iy * g.w + ix) * g.cin + part * 8) * 2) : OOB;
    }
  }
  unsigned b_goff[BSLOTS];
  int b_loff[BSLOTS];
#pragma unroll
  for (int s = 0; s < BSLOTS; ++s) {
    const int v = tid + s * 256;
    const int row = v / (NT * VPP), rem = v % (NT * VPP);
    const int tap = rem / VPP, part = rem % VPP;
    b_goff[s] = (v < BVEC) ? (unsigned)((row * wrow + tap * g.cin_pad + part * 8) * 2) : OOB;
    b_loff[s] = row * RS_B + (tap * KC + part * 8) * 2;
  }

  // ---- this lane's operand addresses
  const int kgrp = lane >> 5, l31 = lane & 31;
  // B operand (pixels): sub-tile mt of wave wid = rows (wid*MT + mt)*2 + (l31 >> 4), col l31 & 15
  const int a_base = ((wid * MT) * 2 + (l31 >> 4)) * ROW_A + (l31 & 15) * PS_A + kgrp * 16;
  const int b_base = l31 * RS_B + kgrp * 16;

  f32x16 acc[MT][NTILE];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int i = 0; i < NTILE; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[m][i][j] = 0.f;

  // NOTE: a cin that is not a multiple of 16 (the 264-channel minibatch-stddev tensor) makes the last
  // chunk read 8 channels of the NEXT pixel; the weight pack is zero there, so they contribute 0.
  bf16x8 ra[ASLOTS], rb[BSLOTS];
  unsigned rs[UNPOOL ? ASLOTS : 1];      // UNPOOL: the sign byte of each staged vector
  bf16x8 rz[UPZ ? ASLOTS : 1];           // UPZ: ... or the eight activations it is taken from
  // UPBWD skip blocks: iteration `it` of the K loop is chunk it % nch of source it / nch
  const int nch = g.cin_pad / KC;
  auto load_chunk = [&](int it) {
    const int c0 = UPBWD ? (it % nch) * KC : it * KC;
    if constexpr (UPBWD) {
      const bool dead = skip_blk && nsrc == 0;      // nobody read this skip image: zeros
      const int simg = skip_blk ? upbwd_source_image(g, img, srcpk, it / nch) : img;
      const __amdgpu_buffer_rsrc_t rs = make_rsrc(x + (size_t)simg * img_elems, dead ? 0u : (unsigned)(img_elems * 2));
#pragma unroll
      for (int s = 0; s < ASLOTS; ++s) ra[s] = buf_load16(rs, a_goff[s] + (unsigned)(c0 * 2));
    } else if constexpr (UPCAT) {
      if (c0 < g.c0) {      // uniform: a chunk lies entirely in one source (c0 % KC == 0)
#pragma unroll
        for (int s = 0; s < ASLOTS; ++s) ra[s] = buf_load16(rx, a_goff[s] + (unsigned)(c0 * 2));
      } else {
#pragma unroll
        for (int s = 0; s < ASLOTS; ++s) ra[s] = buf_load16(rx1, a_goff1[s] + (unsigned)((c0 - g.c0) * 2));
      }
    } else if constexpr (UNPOOL) {
#pragma unroll
      for (int s = 0; s < ASLOTS; ++s) {
        ra[s] = buf_load16(rx, a_goff[s] + (unsigned)(c0 * 2));
        if constexpr (UPZ) rz[s] = buf_load16(rsg, a_goff1[s] + (unsigned)(c0 * 2));
        else rs[s] = buf_load_u8(rsg, a_goff1[s] + (unsigned)(c0 >> 3));
      }
    } else {
#pragma unroll
      for (int s = 0; s < ASLOTS; ++s) ra[s] = buf_load16(rx, a_goff[s] + (unsigned)(c0 * 2));
    }
#pragma unroll
    for (int s = 0; s < BSLOTS; ++s) rb[s] = buf_load16(rw, b_goff[s] + (unsigned)(c0 * 2));
  };
  const size_t full_img = (size_t)g.h * g.w * g.cin;      // UNPOOL: the (never read) full-resolution gradient image
  const bool wr_through = UNPOOL && g.up_store != nullptr && n0 == 0;
  const __amdgpu_buffer_rsrc_t rst = make_rsrc(wr_through ? g.up_store + (size_t)img * full_img : (bf16*)x,
                                               wr_through ? (unsigned)(full_img * 2) : 0u);
  auto store_chunk = [&](int it_st) {
    const int cst = it_st * KC;      // first channel of the chunk being stored
    if constexpr (UNPOOL) {      // AvgPoolGrad + LeakyReluGrad on the way into LDS (border slots: 0 stays 0)
#pragma unroll
      for (int s = 0; s < ASLOTS; ++s) ra[s] = unpool8<F16>(ra[s], UPZ ? sign_bits8(rz[s]) : rs[s], g.up_alpha);
      if (wr_through) {      // uniform
#pragma unroll
        for (int s = 0; s < ASLOTS; ++s)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ra[s]), rst, a_soff[s] + (unsigned)(cst * 2), 0, TG_STORE_AUX);
      }
    }
#pragma unroll
    for (int s = 0; s < ASLOTS; ++s)
      if (s < ASLOTS - 1 || tid + s * 256 < AVEC) *reinterpret_cast<bf16x8*>(sA + a_loff[s]) = ra[s];
#pragma unroll
    for (int s = 0; s < BSLOTS; ++s)
      if (s < BSLOTS - 1 || tid + s * 256 < BVEC) *reinterpret_cast<bf16x8*>(sB + b_loff[s]) = rb[s];
  };

  const int nit = (UPBWD && skip_blk) ? (nsrc > 0 ? nsrc : 1) * nch : nch;
  load_chunk(0);
  for (int it = 0; it < nit; ++it) {
    if (it) __syncthreads();             // everyone is done reading the previous chunk
    store_chunk(it);
    __syncthreads();
    if (it + 1 < nit) load_chunk(it + 1);     // in flight during the MFMAs below
    // K steps of this chunk: (tap, 16-channel half).  The fragments of step s+1 are read from LDS before the MFMAs of
    // step s are issued (two fragment sets; the scheduling barriers keep the compiler from moving the reads back next
    // to their use), so an MFMA never waits a full LDS round trip for its operands.
    constexpr int NSTEP = NT * (KC / 16);
    bf16x8 xf[2][MT], wf[2][NTILE];
    auto read_step = [&](int st, int buf) __attribute__((always_inline)) {
      const int tap = st / (KC / 16), kk = st % (KC / 16);
      const int ky = tap / KW, kx = tap % KW;
#pragma unroll
      for (int m = 0; m < MT; ++m)
        xf[buf][m] = *reinterpret_cast<const bf16x8*>(sA + a_base + (m * 2 + ky) * ROW_A + kx * PS_A + kk * 32);
#pragma unroll
      for (int nt = 0; nt < NTILE; ++nt)
        wf[buf][nt] = *reinterpret_cast<const bf16x8*>(sB + b_base + nt * 32 * RS_B + (tap * KC + kk * 16) * 2);
    };
    read_step(0, 0);
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
      if (st + 1 < NSTEP) read_step(st + 1, (st + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nt = 0; nt < NTILE; ++nt)
#pragma unroll
        for (int m = 0; m < MT; ++m)
          acc[m][nt] = mfma_32x32x16<F16>(wf[st & 1][nt], xf[st & 1][m], acc[m][nt]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue.  acc[m][nt][4q + j] = channel n0 + nt*32 + 8q + 4*kgrp + j of this lane's pixel.
  // After the half-wave swap the low lane (kgrp 0) owns channels [0,16) of the 32-block, the high lane
  // [16,32), each as two 16-byte vectors.
  if constexpr (UPBWD) {
    // up block: the 2x2 sum of the accumulators -> up_out[img, oy/2, ox/2, n0 ..); skip block: the accumulators (already
    // summed over the sources) -> skip_out[img, oy, ox, n0 - c0 ..).  One rounding, of the fp32 sums.
    const int cs = skip_blk ? g.cout - g.c0 : g.c0;      // channels per pixel of the tensor written
    const int chb = skip_blk ? n0 - g.c0 : n0;
    const size_t oimg = skip_blk ? (size_t)g.h * g.w * cs : (size_t)(g.h / 2) * (g.w / 2) * cs;
    bf16* const optr = skip_blk ? g.skip_out : g.up_out;      // NULL: that gradient is not wanted (stores dropped)
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(optr ? optr + (size_t)img * oimg : (bf16*)x, optr ? (unsigned)(oimg * 2) : 0u);
#pragma unroll
    for (int nt = 0; nt < NTILE; ++nt) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int oy = oy0 + (wid * MT + m) * 2 + (l31 >> 4), ox = ox0 + (l31 & 15);
        unsigned p[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = skip_blk ? acc[m][nt][q * 4 + j] : sum_quad(acc[m][nt][q * 4 + j]);
          p[q][0] = pack16x2<F16>(v[0], v[1]);
          p[q][1] = pack16x2<F16>(v[2], v[3]);
        }
        u32x4 o0, o1;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          auto r02 = __builtin_amdgcn_permlane32_swap(p[0][d], p[2][d], false, false);
          auto r13 = __builtin_amdgcn_permlane32_swap(p[1][d], p[3][d], false, false);
          o0[d] = r02[0];
          o0[2 + d] = r02[1];
          o1[d] = r13[0];
          o1[2 + d] = r13[1];
        }
        const int ch0 = chb + nt * 32 + kgrp * 16;
        const bool owner = skip_blk || (l31 & 17) == 0;      // up: the even column of the sub-tile's first row stores
        const unsigned off = skip_blk ? (unsigned)(((oy * g.w + ox) * cs + ch0) * 2)
                                      : (unsigned)((((oy >> 1) * (g.w >> 1) + (ox >> 1)) * cs + ch0) * 2);
        __builtin_amdgcn_raw_buffer_store_b128(o0, ro, (owner && ch0 + 8 <= cs) ? off : OOB, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(o1, ro, (owner && ch0 + 16 <= cs) ? off + 16 : OOB, 0, 0);
      }
    }
    return;
  }
  const size_t out_img = (size_t)g.h * g.w * g.cout;
  // POOL with a sign-mask output: y is not written (a zero-sized resource drops the stores)
  const bool y_dropped = POOL && g.ymask != nullptr;
  const __amdgpu_buffer_rsrc_t ry = make_rsrc(y_dropped ? (const bf16*)g.ypool : y + (size_t)img * out_img,
                                              y_dropped ? 0u : (unsigned)(out_img * 2));
  const __amdgpu_buffer_rsrc_t rmaskout =
      make_rsrc(y_dropped ? g.ymask + (size_t)img * (out_img >> 3) : (unsigned char*)g.ypool, y_dropped ? (unsigned)(out_img >> 3) : 0u);
  const __amdgpu_buffer_rsrc_t rmask =
      make_rsrc(g.mask ? g.mask + (size_t)img * out_img : y, g.mask ? (unsigned)(out_img * 2) : 0u);
  const __amdgpu_buffer_rsrc_t rbias = make_rsrc(bias, (g.epilogue & TG_EPI_BIAS) ? (unsigned)(g.cout * 4) : 0u);
  const __amdgpu_buffer_rsrc_t rpool =
      make_rsrc(POOL ? g.ypool + (size_t)img * (out_img / 4) : y, POOL ? (unsigned)(out_img / 4 * 2) : 0u);
  float stot[NTILE];
#pragma unroll
  for (int nt = 0; nt < NTILE; ++nt) {
    f32x4 bq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)      // OOB (channel >= cout, or no bias) reads 0
      bq[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                             rbias, (unsigned)((n0 + nt * 32 + q * 8 + kgrp * 4) * 4), 0, 0));
    float sv[STATS ? 32 : 1];
    if constexpr (STATS) {
#pragma unroll
      for (int r = 0; r < 32; ++r) sv[r] = 0.f;
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int oy = oy0 + (wid * MT + m) * 2 + (l31 >> 4), ox = ox0 + (l31 & 15);
      unsigned p[4][2];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float a = acc[m][nt][q * 4 + j] + bq[q][j];
          if (g.epilogue & TG_EPI_LRELU) a = lrelu_f(a, g.alpha);
          v[j] = a;
        }
        if (g.mask) {      // uniform
          const int chq = n0 + nt * 32 + q * 8 + kgrp * 4;
          float f[4];
          mask4(rmask, chq + 4 <= g.cout ? (unsigned)(((oy * g.w + ox) * g.cout + chq) * 2) : OOB, g.alpha, f);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] *= f[j];
        }
        p[q][0] = pack16x2<F16>(v[0], v[1]);
        p[q][1] = pack16x2<F16>(v[2], v[3]);
        if constexpr (STATS) {
          const float r4[4] = {unpack16_lo<F16>(p[q][0]), unpack16_hi<F16>(p[q][0]), unpack16_lo<F16>(p[q][1]), unpack16_hi<F16>(p[q][1])};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            sv[q * 4 + j] += r4[j];
            sv[16 + q * 4 + j] = fmaf(r4[j], r4[j], sv[16 + q * 4 + j]);
          }
        }
      }
      // swap: (p[0], p[2]) and (p[1], p[3]); vdst[hi half] <-> src[lo half]
      u32x4 o0, o1;
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        auto r02 = __builtin_amdgcn_permlane32_swap(p[0][d], p[2][d], false, false);
        auto r13 = __builtin_amdgcn_permlane32_swap(p[1][d], p[3][d], false, false);
        // low lane: r02 = (own q0 [ch 0-3], partner q0 [ch 4-7]); high lane: (partner q2 [16-19], own q2 [20-23])
        o0[d] = r02[0];
        o0[2 + d] = r02[1];
        o1[d] = r13[0];
        o1[2 + d] = r13[1];
      }
      const int ch0 = n0 + nt * 32 + kgrp * 16;
      const unsigned off = (unsigned)(((oy * g.w + ox) * g.cout + ch0) * 2);
      __builtin_amdgcn_raw_buffer_store_b128(o0, ry, (ch0 + 8 <= g.cout) ? off : OOB, 0, TG_STORE_AUX);
      __builtin_amdgcn_raw_buffer_store_b128(o1, ry, (ch0 + 16 <= g.cout) ? off + 16 : OOB, 0, TG_STORE_AUX);
      if constexpr (POOL) {
        if (g.ymask) {      // uniform: the sign bits of this lane's 16 channels (y itself is not stored: ry has size 0)
          const unsigned bits = sign_bits16(o0, o1);
          const unsigned moff = (unsigned)((oy * g.w + ox) * (g.cout >> 3) + (ch0 >> 3));
          if (ch0 + 16 <= g.cout) __builtin_amdgcn_raw_buffer_store_b16((short)bits, rmaskout, moff, 0, 0);
          else if (ch0 + 8 <= g.cout) __builtin_amdgcn_raw_buffer_store_b8((char)(bits & 0xffu), rmaskout, moff, 0, 0);
        }
        unsigned pp[4][2];
        pool_quad<F16>(p, pp);
        u32x4 q0, q1;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          auto r02 = __builtin_amdgcn_permlane32_swap(pp[0][d], pp[2][d], false, false);
          auto r13 = __builtin_amdgcn_permlane32_swap(pp[1][d], pp[3][d], false, false);
          q0[d] = r02[0];
          q0[2 + d] = r02[1];
          q1[d] = r13[0];
          q1[2 + d] = r13[1];
        }
        const bool owner = (l31 & 17) == 0;      // even column of the sub-tile's first row
        const unsigned poff = (unsigned)((((oy >> 1) * (g.w >> 1) + (ox >> 1)) * g.cout + ch0) * 2);
        __builtin_amdgcn_raw_buffer_store_b128(q0, rpool, (owner && ch0 + 8 <= g.cout) ? poff : OOB, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(q1, rpool, (owner && ch0 + 16 <= g.cout) ? poff + 16 : OOB, 0, 0);
      }
    }
    if constexpr (STATS) {
      half_wave_transpose_sum<0, 5>(sv, l31);
      stot[nt] = sv[0];
    }
  }
  if constexpr (STATS)
    stats_flush<BN>(stot, reinterpret_cast<float*>(sA), tid, n0, g.cout,
                    g.stats + ((size_t)img * g.stat_chunks + tile_in_img) * 2 * g.cout);
}


// ------------------------------------------------------------------------------------------------
// Weight-resident variant for the THIN layers (cin_pad <= 64, i.e. NCH <= 2 chunks of 32 channels, or one of
// 16): with K = 9*cin that small, re-staging the workgroup's [BN x 9*cin] weight slice for every 128-pixel
// tile costs more L2->LDS traffic than the activations themselves (16 k tiles x 37 KB at 128x128).  Here a
// workgroup stages its weight slice ONCE and walks `tiles_per_wg` consecutive tiles, so per tile it only moves
// the input halo; the next tile's halo loads are in flight during the MFMAs of the current one.
// ------------------------------------------------------------------------------------------------
// EPI: which of the per-tile epilogue operands exist -- bit 0 the bias row (16 VGPRs per 32-channel block, held across the
// workgroup's tiles), bit 1 the LeakyReLU mask of the masked backward-data (8 VGPRs per block).  Compile-time so that a
// launch without them does not pay their registers: these kernels are short of RESIDENT WORKGROUPS, not of anything else
// (the single-chunk 16-channel variant: 88 VGPRs = 5 workgroups per CU with both, 8 by LDS).
template <int KH, int KC, int BN, int NCH, int MODE = 0, bool F16 = false, int EPI = 3>
__global__ __launch_bounds__(256) void conv_tile_wres_kernel(const bf16* __restrict__ x, const bf16* __restrict__ wp,
                                                             const float* __restrict__ bias, bf16* __restrict__ y,
                                                             const TileGeom g) {
  constexpr bool HAS_BIAS = (EPI & 1) != 0, HAS_MASK = (EPI & 2) != 0;
  // UPBWD: see TileGeom::up_out.  MODE 4 (UPBOTH) is UPBWD with ONE 64-channel block that holds both halves of a 32 + 32
  // concat: channel block 0 = "up" (flushed per source image), block 1 = "skip" (summed over the sources of a skip image)
  // -- the workgroup walks (skip image, source) pairs, so every gy tile is staged ONCE (the two-block form reads gy twice
  // and runs twice the tile iterations, which is what these thin kernels are bound by)
  constexpr bool STATS = MODE == 1, POOL = MODE == 2, UPBOTH = MODE == 4, UPBWD = MODE == 3 || UPBOTH;
  constexpr bool UNPOOL = MODE == 5 || MODE == 6;      // the input tile is built from the pooled gradient + sign bytes (TileGeom::up_src)
  constexpr bool UPZ = MODE == 6;                      // ... the signs taken from the activation tensor itself (TileGeom::up_z)
  static_assert(!UPBOTH || BN == 64, "UPBOTH: one up block + one skip block");
  constexpr int KW = KH, NT = KH * KW;
  constexpr int TW = 16, TH = 8;
  constexpr int HWX = TW + KW - 1, HH = TH + KH - 1;
  constexpr int VPP = KC / 8;
  constexpr int PS_A = HaloLds<KC, HWX>::PS, ROW_A = HaloLds<KC, HWX>::ROW;      // bank-conflict-free fragment reads
  constexpr int RS_B = NT * KC * 2 + 16;
  constexpr int AVEC = HH * HWX * VPP;
  constexpr int ASLOTS = (AVEC + 255) / 256;
  constexpr int BVEC = BN * NT * VPP;
  constexpr int BSLOTS = (BVEC + 255) / 256;
  constexpr int NTILE = BN / 32;
  constexpr int A_BYTES = (HH * ROW_A + 15) & ~15;
  constexpr int B_BYTES = BN * RS_B;

  unsigned char* sA = tile_smem;
  unsigned char* sB = tile_smem + A_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int n0 = blockIdx.y * BN;
  const int wrow = NT * g.cin_pad;
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(wp + (size_t)n0 * wrow, (unsigned)((size_t)BN * wrow * 2));
  // UPBWD: "skip" blocks (output channels >= c0) walk the tiles of the n1 SKIP images, each over its source images
  const bool skip_blk = UPBOTH || (UPBWD && n0 >= g.c0);      // UPBOTH: the walk of a skip block
  if constexpr (UPBWD) {
    int wg0 = blockIdx.x;
    if ((gridDim.x & 7) == 0) wg0 = (wg0 & 7) * (gridDim.x >> 3) + (wg0 >> 3);
    if (skip_blk && wg0 * g.tiles_per_wg >= g.tiles_x * g.tiles_y * g.n1) return;      // block-uniform, before any barrier
  }

  // ---- stage the whole weight slice (all chunks) once
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    bf16x8 rb[BSLOTS];
#pragma unroll
    for (int s = 0; s < BSLOTS; ++s) {
      const int v = tid + s * 256;
      const int row = v / (NT * VPP), rem = v % (NT * VPP);
      const int tap = rem / VPP, part = rem % VPP;
      const unsigned off = (v < BVEC) ? (unsigned)((row * wrow + tap * g.cin_pad + ch * KC + part * 8) * 2) : OOB;
      rb[s] = buf_load16(rw, off);
    }
#pragma unroll
    for (int s = 0; s < BSLOTS; ++s) {
      const int v = tid + s * 256;
      const int row = v / (NT * VPP), rem = v % (NT * VPP);
      const int tap = rem / VPP, part = rem % VPP;
      if (s < BSLOTS - 1 || v < BVEC)
        *reinterpret_cast<bf16x8*>(sB + ch * B_BYTES + row * RS_B + (tap * KC + part * 8) * 2) = rb[s];
    }
  }

  // ---- tile-independent slot geometry
  int a_hy[ASLOTS], a_hx[ASLOTS], a_loff[ASLOTS];
#pragma unroll
  for (int s = 0; s < ASLOTS; ++s) {
    const int v = tid + s * 256;
    const int px = v / VPP, part = v % VPP;
    a_hy[s] = (v < AVEC) ? px / HWX : -100000;      // unused slot: never in range
    a_hx[s] = px % HWX;
    a_loff[s] = (px / HWX) * ROW_A + (px % HWX) * PS_A + part * 16;
  }
  const int kgrp = lane >> 5, l31 = lane & 31;
  const int a_base = (wid * 2 + (l31 >> 4)) * ROW_A + (l31 & 15) * PS_A + kgrp * 16;
  const int b_base = l31 * RS_B + kgrp * 16;

  // XCD-aware order of workgroups, then consecutive tiles inside a workgroup
  int wg = blockIdx.x;
  const int nwg = gridDim.x;
  if ((nwg & 7) == 0) wg = (wg & 7) * (nwg >> 3) + (wg >> 3);
  const int t_begin = wg * g.tiles_per_wg;
  int t_end = t_begin + g.tiles_per_wg;
  if (t_end > g.nblk) t_end = g.nblk;
  if constexpr (UPBWD) {
    if (skip_blk && t_end > g.tiles_x * g.tiles_y * g.n1) t_end = g.tiles_x * g.tiles_y * g.n1;
  }
  const size_t img_elems = (size_t)g.h * g.w * g.cin;
  const size_t out_img = (size_t)g.h * g.w * g.cout;

  static_assert(NCH == 1, "the weight-resident kernel is dispatched for single-chunk layers (cin_pad <= 32)");
  // One register stage (the halo of tile t+1 is in flight while tile t is reduced).  Measured and not kept: two stages
  // (t+1 and t+2 in flight) -- 100 instead of 84 VGPRs = 4 instead of 5 workgroups per CU, 70 vs 65 us on 256x256x16
  // n64 in an A/B on one box.  A pure copy with the same 8x16-tile + halo access pattern and 8 workgroups per CU
  // reaches 5.2-5.7 TB/s (tools/probes/tile_copy.hip; a flat copy 6.8): resident workgroups, not prefetch depth or the
  // 2-D pattern, are what this kernel is short of.
  struct Stage {
    bf16x8 ra[ASLOTS];
    unsigned rs[UNPOOL ? ASLOTS : 1];      // UNPOOL: the sign byte of each staged vector
    bf16x8 rz[UPZ ? ASLOTS : 1];           // UPZ: ... or the eight activations it is taken from
  };
  // src (UPBWD skip blocks): which of the tile's source images
  auto load_a = [&](Stage& st, int t, int src = 0) __attribute__((always_inline)) {
    unsigned live = t < t_end;
    if (!live) t = t_begin;
    const int tx = t % g.tiles_x;
    const int r = t / g.tiles_x;
    const int ty = r % g.tiles_y;
    int img = r / g.tiles_y;
    if constexpr (UPBWD) {
      if (skip_blk) {
        unsigned pk;
        const int ns = upbwd_sources(g, img, &pk);
        if (ns == 0) live = 0;      // nobody read this skip image: the tile is all zeros
        img = upbwd_source_image(g, img, pk, src);
      }
    }
    if constexpr (UNPOOL) {
      const size_t pool_elems = (size_t)(g.h / 2) * (g.w / 2) * g.cin, sign_bytes = (size_t)g.h * g.w * (g.cin >> 3);
      const __amdgpu_buffer_rsrc_t rp = make_rsrc(g.up_src + (size_t)img * pool_elems, (unsigned)(pool_elems * 2));
      const __amdgpu_buffer_rsrc_t rsg =
          UPZ ? make_rsrc(g.up_z + (size_t)img * img_elems, (unsigned)(img_elems * 2))
              : make_rsrc(g.up_signs + (size_t)img * sign_bytes, (unsigned)sign_bytes);
#pragma unroll
      for (int s = 0; s < ASLOTS; ++s) {
        const int iy = ty * TH + a_hy[s] - g.pad, ix = tx * TW + a_hx[s] - g.pad;
        const bool ok = live && iy >= 0 && iy < g.h && ix >= 0 && ix < g.w;
        const int part = (tid + s * 256) % VPP;
        st.ra[s] = buf_load16(rp, ok ? (unsigned)((((iy >> 1) * (g.w >> 1) + (ix >> 1)) * g.cin + part * 8) * 2) : OOB);
        if constexpr (UPZ) st.rz[s] = buf_load16(rsg, ok ? (unsigned)(((iy * g.w + ix) * g.cin + part * 8) * 2) : OOB);
        else st.rs[s] = buf_load_u8(rsg, ok ? (unsigned)((iy * g.w + ix) * (g.cin >> 3) + part) : OOB);
      }
      return;
    }
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x + (size_t)img * img_elems, (unsigned)(img_elems * 2));
#pragma unroll
    for (int s = 0; s < ASLOTS; ++s) {
      const int iy = ty * TH + a_hy[s] - g.pad, ix = tx * TW + a_hx[s] - g.pad;
      const bool ok = live && iy >= 0 && iy < g.h && ix >= 0 && ix < g.w;
      const int part8 = ((tid + s * 256) % VPP) * 8;
      st.ra[s] = buf_load16(rx, ok ? (unsigned)(((iy * g.w + ix) * g.cin + part8) * 2) : OOB);
    }
  };

  const __amdgpu_buffer_rsrc_t rbias = make_rsrc(bias, (HAS_BIAS && (g.epilogue & TG_EPI_BIAS)) ? (unsigned)(g.cout * 4) : 0u);
  f32x4 bq[HAS_BIAS ? NTILE : 1][4];
  if constexpr (HAS_BIAS) {
#pragma unroll
    for (int nt = 0; nt < NTILE; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        bq[nt][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                   rbias, (unsigned)((n0 + nt * 32 + q * 8 + kgrp * 4) * 4), 0, 0));
  }

  // STATS: a workgroup's tiles belong to ONE image (the launcher picks tiles_per_wg as a divisor of the tiles per image).
  // Per tile each lane folds its 32 values per channel block (16 channels, value and square) by the first ST steps of
  // the transposing half-wave sum and accumulates the 32 >> ST that are left; the remaining steps run once at the end.
  // One 32-channel block: ST = 0 -- everything stays in the lane until the end (16 packed adds / FMAs per tile, 32
  // accumulators: 4 instead of 5 workgroups per CU).  Two blocks: ST = 2 -- the quad steps run per tile on DPP lane
  // exchanges and 8 accumulators per block are kept.
  constexpr int ST = NTILE == 1 ? 0 : 2;
  float sacc[NTILE][STATS ? (32 >> ST) : 1];
  if constexpr (STATS) {
#pragma unroll
    for (int nt = 0; nt < NTILE; ++nt)
#pragma unroll
      for (int i = 0; i < (32 >> ST); ++i) sacc[nt][i] = 0.f;
  }

  bool first = true;
  f32x16 acc[NTILE];      // outside the per-tile lambda: under UPBWD the accumulators live across the source images of a tile
  // one tile: stage -> LDS, refill the stage with tile t + 1, MFMAs, epilogue
  // UPBWD: (t, src) of n_src is reduced now, (tn, sn) is the next one to stage; the epilogue runs after the last source
  auto process = [&](Stage& st, int t, int src = 0, int n_src = 1, int tn = 0, int sn = 0) __attribute__((always_inline)) {
    const int tx = t % g.tiles_x;
    const int r = t / g.tiles_x;
    const int ty = r % g.tiles_y;
    const int img = r / g.tiles_y;
    const int oy = ty * TH + wid * 2 + (l31 >> 4), ox = tx * TW + (l31 & 15);
    const bool y_dropped = POOL && g.ymask != nullptr;      // sign-mask output: y is not written
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(y_dropped ? (const bf16*)g.ypool : y + (size_t)img * out_img,
                                                y_dropped ? 0u : (unsigned)(out_img * 2));
    const __amdgpu_buffer_rsrc_t rmaskout =
        make_rsrc(y_dropped ? g.ymask + (size_t)img * (out_img >> 3) : (unsigned char*)g.ypool, y_dropped ? (unsigned)(out_img >> 3) : 0u);
    const __amdgpu_buffer_rsrc_t rmask =
        make_rsrc(g.mask ? g.mask + (size_t)img * out_img : y, g.mask ? (unsigned)(out_img * 2) : 0u);
    if constexpr (UNPOOL) {      // AvgPoolGrad + LeakyReluGrad on the way into LDS (border / dead slots: 0 stays 0)
#pragma unroll
      for (int s = 0; s < ASLOTS; ++s) st.ra[s] = unpool8<F16>(st.ra[s], UPZ ? sign_bits8(st.rz[s]) : st.rs[s], g.up_alpha);
      if (g.up_store && n0 == 0) {      // uniform: the interior pixels of this tile also go to the gradient tensor itself
        const __amdgpu_buffer_rsrc_t rst = make_rsrc(g.up_store + (size_t)img * img_elems, (unsigned)(img_elems * 2));
#pragma unroll
        for (int s = 0; s < ASLOTS; ++s) {
          const int iy = ty * TH + a_hy[s] - g.pad, ix = tx * TW + a_hx[s] - g.pad;
          const bool own = a_hy[s] >= g.pad && a_hy[s] < g.pad + TH && a_hx[s] >= g.pad && a_hx[s] < g.pad + TW;
          const int part8 = ((tid + s * 256) % VPP) * 8;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, st.ra[s]), rst,
                                                 own ? (unsigned)(((iy * g.w + ix) * g.cin + part8) * 2) : OOB, 0, TG_STORE_AUX);
        }
      }
    }
    if (!first) __syncthreads();          // everyone finished reading the previous halo
    first = false;
#pragma unroll
    for (int s = 0; s < ASLOTS; ++s)
      if (s < ASLOTS - 1 || tid + s * 256 < AVEC) *reinterpret_cast<bf16x8*>(sA + a_loff[s]) = st.ra[s];
    __syncthreads();                      // (the first one also covers the weight staging)
    if constexpr (UPBWD) load_a(st, tn, sn);
    else load_a(st, t + 1);
    // the LeakyReLU mask of the epilogue (masked backward-data) is requested NOW, so that it lands during the MFMAs
    u32x2 zm[HAS_MASK ? NTILE : 1][4];
    if (HAS_MASK && g.mask) {      // uniform
#pragma unroll
      for (int nt = 0; nt < NTILE; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int chq = n0 + nt * 32 + q * 8 + kgrp * 4;
          zm[nt][q] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(
              rmask, chq + 4 <= g.cout ? (unsigned)(((oy * g.w + ox) * g.cout + chq) * 2) : OOB, 0, 0));
        }
    }
    if (!UPBWD || src == 0) {
#pragma unroll
      for (int i = 0; i < NTILE; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    } else if (UPBOTH) {      // the up block restarts with every source image
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[0][j] = 0.f;
    }
#pragma unroll
    for (int ky = 0; ky < KH; ++ky) {
#pragma unroll
      for (int kx = 0; kx < KW; ++kx) {
#pragma unroll
        for (int kk = 0; kk < KC / 16; ++kk) {
          const bf16x8 xf = *reinterpret_cast<const bf16x8*>(sA + a_base + ky * ROW_A + kx * PS_A + kk * 32);
#pragma unroll
          for (int nt = 0; nt < NTILE; ++nt) {
            const bf16x8 wf = *reinterpret_cast<const bf16x8*>(sB + b_base + nt * 32 * RS_B + ((ky * KW + kx) * KC + kk * 16) * 2);
            acc[nt] = mfma_32x32x16<F16>(wf, xf, acc[nt]);
          }
        }
      }
    }
    // ---- epilogue of tile t
    if constexpr (UPBOTH) {
      // block 0 -> the 2x2 sums of THIS source image's up gradient; block 1 -> the skip gradient after the last source
      unsigned pk;
      const int ns_here = upbwd_sources(g, img, &pk);
      const int simg = upbwd_source_image(g, img, pk, src);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const bool up = nt == 0;
        if (!up && src + 1 < n_src) continue;
        const int cs = up ? g.c0 : g.cout - g.c0;
        const size_t oimg = up ? (size_t)(g.h / 2) * (g.w / 2) * cs : (size_t)g.h * g.w * cs;
        bf16* const optr = up ? g.up_out : g.skip_out;
        const bool wanted = optr != nullptr && !(up && ns_here == 0);      // an unread skip image has no source image
        const __amdgpu_buffer_rsrc_t ro =
            make_rsrc(wanted ? optr + (size_t)(up ? simg : img) * oimg : (bf16*)x, wanted ? (unsigned)(oimg * 2) : 0u);
        unsigned p[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = up ? sum_quad(acc[nt][q * 4 + j]) : acc[nt][q * 4 + j];
          p[q][0] = pack16x2<F16>(v[0], v[1]);
          p[q][1] = pack16x2<F16>(v[2], v[3]);
        }
        u32x4 o0, o1;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          auto r02 = __builtin_amdgcn_permlane32_swap(p[0][d], p[2][d], false, false);
          auto r13 = __builtin_amdgcn_permlane32_swap(p[1][d], p[3][d], false, false);
          o0[d] = r02[0];
          o0[2 + d] = r02[1];
          o1[d] = r13[0];
          o1[2 + d] = r13[1];
        }
        const int ch0 = kgrp * 16;
        const bool owner = !up || (l31 & 17) == 0;
        const unsigned off = up ? (unsigned)((((oy >> 1) * (g.w >> 1) + (ox >> 1)) * cs + ch0) * 2)
                                : (unsigned)(((oy * g.w + ox) * cs + ch0) * 2);
        __builtin_amdgcn_raw_buffer_store_b128(o0, ro, (owner && ch0 + 8 <= cs) ? off : OOB, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(o1, ro, (owner && ch0 + 16 <= cs) ? off + 16 : OOB, 0, 0);
      }
      return;
    } else if constexpr (UPBWD) {      // as in conv_tile_kernel
      if (src + 1 < n_src) return;
      const int cs = skip_blk ? g.cout - g.c0 : g.c0;
      const int chb = skip_blk ? n0 - g.c0 : n0;
      const size_t oimg = skip_blk ? (size_t)g.h * g.w * cs : (size_t)(g.h / 2) * (g.w / 2) * cs;
      bf16* const optr = skip_blk ? g.skip_out : g.up_out;      // NULL: that gradient is not wanted (stores dropped)
      const __amdgpu_buffer_rsrc_t ro = make_rsrc(optr ? optr + (size_t)img * oimg : (bf16*)x, optr ? (unsigned)(oimg * 2) : 0u);
#pragma unroll
      for (int nt = 0; nt < NTILE; ++nt) {
        unsigned p[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = skip_blk ? acc[nt][q * 4 + j] : sum_quad(acc[nt][q * 4 + j]);
          p[q][0] = pack16x2<F16>(v[0], v[1]);
          p[q][1] = pack16x2<F16>(v[2], v[3]);
        }
        u32x4 o0, o1;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          auto r02 = __builtin_amdgcn_permlane32_swap(p[0][d], p[2][d], false, false);
          auto r13 = __builtin_amdgcn_permlane32_swap(p[1][d], p[3][d], false, false);
          o0[d] = r02[0];
          o0[2 + d] = r02[1];
          o1[d] = r13[0];
          o1[2 + d] = r13[1];
        }
        const int ch0 = chb + nt * 32 + kgrp * 16;
        const bool owner = skip_blk || (l31 & 17) == 0;
        const unsigned off = skip_blk ? (unsigned)(((oy * g.w + ox) * cs + ch0) * 2)
                                      : (unsigned)((((oy >> 1) * (g.w >> 1) + (ox >> 1)) * cs + ch0) * 2);
        __builtin_amdgcn_raw_buffer_store_b128(o0, ro, (owner && ch0 + 8 <= cs) ? off : OOB, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(o1, ro, (owner && ch0 + 16 <= cs) ? off + 16 : OOB, 0, 0);
      }
      return;
    }
#pragma unroll
    for (int nt = 0; nt < NTILE; ++nt) {
      unsigned p[4][2];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float a = acc[nt][q * 4 + j];
          if constexpr (HAS_BIAS) a += bq[nt][q][j];
          if (g.epilogue & TG_EPI_LRELU) a = lrelu_f(a, g.alpha);
          v[j] = a;
        }
        if (HAS_MASK && g.mask) {      // uniform: a positive bf16 is a positive int16 pattern
          const u32x2 z = zm[HAS_MASK ? nt : 0][q];
          v[0] *= (short)(z[0] & 0xffffu) > 0 ? 1.f : g.alpha;
          v[1] *= (short)(z[0] >> 16) > 0 ? 1.f : g.alpha;
          v[2] *= (short)(z[1] & 0xffffu) > 0 ? 1.f : g.alpha;
          v[3] *= (short)(z[1] >> 16) > 0 ? 1.f : g.alpha;
        }
        p[q][0] = pack16x2<F16>(v[0], v[1]);
        p[q][1] = pack16x2<F16>(v[2], v[3]);
        if constexpr (STATS) {
          const float r4[4] = {unpack16_lo<F16>(p[q][0]), unpack16_hi<F16>(p[q][0]), unpack16_lo<F16>(p[q][1]), unpack16_hi<F16>(p[q][1])};
          if constexpr (ST == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              sacc[nt][q * 4 + j] += r4[j];
              sacc[nt][16 + q * 4 + j] = fmaf(r4[j], r4[j], sacc[nt][16 + q * 4 + j]);
            }
          } else {
            sacc[nt][q] += quad_fold4(r4[0], r4[1], r4[2], r4[3], l31);
            sacc[nt][4 + q] += quad_fold4(r4[0] * r4[0], r4[1] * r4[1], r4[2] * r4[2], r4[3] * r4[3], l31);
          }
        }
      }
      u32x4 o0, o1;
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        auto r02 = __builtin_amdgcn_permlane32_swap(p[0][d], p[2][d], false, false);
        auto r13 = __builtin_amdgcn_permlane32_swap(p[1][d], p[3][d], false, false);
        o0[d] = r02[0];
        o0[2 + d] = r02[1];
        o1[d] = r13[0];
        o1[2 + d] = r13[1];
      }
      const int ch0 = n0 + nt * 32 + kgrp * 16;
      const unsigned off = (unsigned)(((oy * g.w + ox) * g.cout + ch0) * 2);
      __builtin_amdgcn_raw_buffer_store_b128(o0, ry, (ch0 + 8 <= g.cout) ? off : OOB, 0, TG_STORE_AUX);
      __builtin_amdgcn_raw_buffer_store_b128(o1, ry, (ch0 + 16 <= g.cout) ? off + 16 : OOB, 0, TG_STORE_AUX);
      if constexpr (POOL) {
        if (g.ymask) {      // uniform: the sign bits of this lane's 16 channels (y itself is not stored: ry has size 0)
          const unsigned bits = sign_bits16(o0, o1);
          const unsigned moff = (unsigned)((oy * g.w + ox) * (g.cout >> 3) + (ch0 >> 3));
          if (ch0 + 16 <= g.cout) __builtin_amdgcn_raw_buffer_store_b16((short)bits, rmaskout, moff, 0, 0);
          else if (ch0 + 8 <= g.cout) __builtin_amdgcn_raw_buffer_store_b8((char)(bits & 0xffu), rmaskout, moff, 0, 0);
        }
        unsigned pp[4][2];
        pool_quad<F16>(p, pp);
        u32x4 q0, q1;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          auto r02 = __builtin_amdgcn_permlane32_swap(pp[0][d], pp[2][d], false, false);
          auto r13 = __builtin_amdgcn_permlane32_swap(pp[1][d], pp[3][d], false, false);
          q0[d] = r02[0];
          q0[2 + d] = r02[1];
          q1[d] = r13[0];
          q1[2 + d] = r13[1];
        }
        const __amdgpu_buffer_rsrc_t rpool = make_rsrc(g.ypool + (size_t)img * (out_img / 4), (unsigned)(out_img / 4 * 2));
        const bool owner = (l31 & 17) == 0;
        const unsigned poff = (unsigned)((((oy >> 1) * (g.w >> 1) + (ox >> 1)) * g.cout + ch0) * 2);
        __builtin_amdgcn_raw_buffer_store_b128(q0, rpool, (owner && ch0 + 8 <= g.cout) ? poff : OOB, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(q1, rpool, (owner && ch0 + 16 <= g.cout) ? poff + 16 : OOB, 0, 0);
      }
    }
  };

  Stage sa;
  if constexpr (UPBWD) {
    auto n_sources = [&](int t) __attribute__((always_inline)) {
      if (!skip_blk || t >= t_end) return 1;
      unsigned pk;
      const int ns = upbwd_sources(g, t / (g.tiles_x * g.tiles_y), &pk);
      return ns > 0 ? ns : 1;      // an unread skip image is one all-zero "source"
    };
    int t = t_begin, src = 0, ns = n_sources(t);
    load_a(sa, t, 0);
    while (t < t_end) {
      int tn = t, sn = src + 1;
      if (sn >= ns) {
        tn = t + 1;
        sn = 0;
      }
      process(sa, t, src, ns, tn, sn);
      if (tn != t) ns = n_sources(tn);
      t = tn;
      src = sn;
    }
    return;
  }
  load_a(sa, t_begin);
  for (int t = t_begin; t < t_end; ++t) process(sa, t);
  if constexpr (STATS) {
    float stot[NTILE];
#pragma unroll
    for (int nt = 0; nt < NTILE; ++nt) {
      half_wave_transpose_sum<ST, 5>(sacc[nt], l31);
      stot[nt] = sacc[nt][0];
    }
    const int tpi = g.tiles_x * g.tiles_y;
    stats_flush<BN>(stot, reinterpret_cast<float*>(sA), tid, n0, g.cout,
                    g.stats + ((size_t)(t_begin / tpi) * g.stat_chunks + (t_begin % tpi) / g.tiles_per_wg) * 2 * g.cout);
  }
}

// ------------------------------------------------------------------------------------------------
// Thin-OUTPUT variant: 3x3 layers with <= 16 output channels (the 256x256 stage: 16 -> 16 forward with or without the
// statistics epilogue, the discriminators' first conv, their masked backward-data).  In the 32-wide blocks above half of
// every MFMA, of every fragment read and of the epilogue is padding there.  Here M = the 16 output channels of
// v_mfma_f32_16x16x32:
//   workgroup = 4 waves, tile = 8 rows x 16 cols as above; wave w owns rows 2w, 2w + 1 = two 16-pixel blocks (N = 16);
//   K = 32 is a PAIR of taps of a 16-channel layer (lane group q = lane / 16: taps (t0, t0, t1, t1), channel halves
//   (0, 1, 0, 1); the ninth tap pairs with zero weights) or ONE tap of a 32-channel layer (q = its 8-channel quarter);
//   the weights -- 5 or 9 A fragments, <= 36 VGPRs -- are loaded from the pack ONCE per workgroup and stay in registers: LDS
//   holds the pixel halo only and is read 10 (18) times 1 KB per wave and tile instead of 18 (36);
//   accumulators: 4 channels (4q .. 4q+3) of the lane's pixel per block -> 8-byte NHWC stores, four lane groups = the
//   pixel's 32 bytes.
// MODE 0: plain (bias / LeakyReLU / mask epilogue), 1: + statistics partials of the rounded outputs (layout of stats_flush).
// ------------------------------------------------------------------------------------------------
template <int KC, int MODE, bool F16, int EPI>
__global__ __launch_bounds__(256) void conv_thin16_kernel(const bf16* __restrict__ x, const bf16* __restrict__ wp,
                                                          const float* __restrict__ bias, bf16* __restrict__ y,
                                                          const TileGeom g) {
  static_assert(KC == 16 || KC == 32, "one 16- or 32-channel chunk");
  constexpr bool STATS = MODE == 1, HAS_BIAS = (EPI & 1) != 0, HAS_MASK = (EPI & 2) != 0;
  constexpr int NT = 9, TW = 16, TH = 8, HWX = TW + 2, HH = TH + 2;
  constexpr int VPP = KC / 8, PS_A = THIN_PS;      // bank-conflict-free fragment reads: see HaloLds / THIN_PS
  constexpr int AVEC = HH * HWX * VPP, ASLOTS = (AVEC + 255) / 256;
  constexpr int NP = KC == 16 ? 5 : 9;      // MFMAs (K = 32) per pixel block: tap pairs / taps
  unsigned char* sA = tile_smem;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int q = lane >> 4, c16 = lane & 15;

  // ---- the weights: A fragments, row = output channel c16, k = this lane group's 8 channels of its tap
  const int wrow = NT * g.cin_pad;
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(wp, (unsigned)((size_t)g.cout * wrow * 2));
  bf16x8 wf[NP];
  int a_off[NP];      // LDS byte offset of this lane's pixel fragment of MFMA p, relative to its pixel block's origin
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int tap = KC == 16 ? 2 * p + (q >> 1) : p;
    const int part = KC == 16 ? (q & 1) : q;
    const bool real = tap < NT && c16 < g.cout;
    wf[p] = buf_load16(rw, real ? (unsigned)((c16 * wrow + tap * g.cin_pad + part * 8) * 2) : OOB);
    const int tp = tap < NT ? tap : NT - 1;      // the zero-weight half reads a valid address
    a_off[p] = ((tp / 3) * HWX + (tp % 3)) * PS_A + part * 16;
  }
  const int a_base = ((wid * 2) * HWX + c16) * PS_A;      // + pb * HWX * PS_A + a_off[p]

  // ---- tile-independent staging geometry (as conv_tile_wres_kernel)
  int a_hy[ASLOTS], a_hx[ASLOTS], a_loff[ASLOTS];
#pragma unroll
  for (int s = 0; s < ASLOTS; ++s) {
    const int v = tid + s * 256;
    const int px = v / VPP, part = v % VPP;
    a_hy[s] = (v < AVEC) ? px / HWX : -100000;
    a_hx[s] = px % HWX;
    a_loff[s] = px * PS_A + part * 16;
  }
  int wg = blockIdx.x;
  const int nwg = gridDim.x;
  if ((nwg & 7) == 0) wg = (wg & 7) * (nwg >> 3) + (wg >> 3);
  const int t_begin = wg * g.tiles_per_wg;
  int t_end = t_begin + g.tiles_per_wg;
  if (t_end > g.nblk) t_end = g.nblk;
  const size_t img_elems = (size_t)g.h * g.w * g.cin;
  const size_t out_img = (size_t)g.h * g.w * g.cout;

  struct Stage {
    bf16x8 ra[ASLOTS];
  };
  auto load_a = [&](Stage& st, int t) __attribute__((always_inline)) {
    const bool live = t < t_end;
    if (!live) t = t_begin;
    const int tx = t % g.tiles_x;
    const int r = t / g.tiles_x;
    const int ty = r % g.tiles_y;
    const int img = r / g.tiles_y;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x + (size_t)img * img_elems, (unsigned)(img_elems * 2));
#pragma unroll
    for (int s = 0; s < ASLOTS; ++s) {
      const int iy = ty * TH + a_hy[s] - g.pad, ix = tx * TW + a_hx[s] - g.pad;
      const bool ok = live && iy >= 0 && iy < g.h && ix >= 0 && ix < g.w;
      const int part8 = ((tid + s * 256) % VPP) * 8;
      st.ra[s] = buf_load16(rx, ok ? (unsigned)(((iy * g.w + ix) * g.cin + part8) * 2) : OOB);
    }
  };

  const __amdgpu_buffer_rsrc_t rbias = make_rsrc(bias, (HAS_BIAS && (g.epilogue & TG_EPI_BIAS)) ? (unsigned)(g.cout * 4) : 0u);
  f32x4 bq = {0.f, 0.f, 0.f, 0.f};
  if constexpr (HAS_BIAS) bq = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rbias, (unsigned)(q * 16), 0, 0));

  float sacc[STATS ? 8 : 1];      // per lane: sums (0..3) and sums of squares (4..7) of its 4 channels over its pixels
  if constexpr (STATS) {
#pragma unroll
    for (int i = 0; i < 8; ++i) sacc[i] = 0.f;
  }

  bool first = true;
  Stage sa;
  load_a(sa, t_begin);
  for (int t = t_begin; t < t_end; ++t) {
    const int tx = t % g.tiles_x;
    const int r = t / g.tiles_x;
    const int ty = r % g.tiles_y;
    const int img = r / g.tiles_y;
    if (!first) __syncthreads();
    first = false;
#pragma unroll
    for (int s = 0; s < ASLOTS; ++s)
      if (s < ASLOTS - 1 || tid + s * 256 < AVEC) *reinterpret_cast<bf16x8*>(sA + a_loff[s]) = sa.ra[s];
    __syncthreads();
    load_a(sa, t + 1);
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(y + (size_t)img * out_img, (unsigned)(out_img * 2));
    const __amdgpu_buffer_rsrc_t rmask =
        make_rsrc(g.mask ? g.mask + (size_t)img * out_img : y, g.mask ? (unsigned)(out_img * 2) : 0u);
    const int ox = tx * TW + c16;
    u32x2 zm[2];
    if (HAS_MASK && g.mask) {      // uniform; requested now, lands during the MFMAs
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        const int oy = ty * TH + wid * 2 + pb;
        zm[pb] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(
            rmask, 4 * q + 4 <= g.cout ? (unsigned)(((oy * g.w + ox) * g.cout + 4 * q) * 2) : OOB, 0, 0));
      }
    }
    f32x4 acc[2];
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) acc[pb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < NP; ++p) {
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(sA + a_base + pb * HWX * PS_A + a_off[p]);
        acc[pb] = mfma_16x16x32<F16>(wf[p], xf, acc[pb]);
      }
    }
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
      const int oy = ty * TH + wid * 2 + pb;
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float a = acc[pb][j];
        if constexpr (HAS_BIAS) a += bq[j];
        if (g.epilogue & TG_EPI_LRELU) a = lrelu_f(a, g.alpha);
        v[j] = a;
      }
      if (HAS_MASK && g.mask) {      // uniform: a positive bf16 / f16 is a positive int16 pattern
        const u32x2 z = zm[pb];
        v[0] *= (short)(z[0] & 0xffffu) > 0 ? 1.f : g.alpha;
        v[1] *= (short)(z[0] >> 16) > 0 ? 1.f : g.alpha;
        v[2] *= (short)(z[1] & 0xffffu) > 0 ? 1.f : g.alpha;
        v[3] *= (short)(z[1] >> 16) > 0 ? 1.f : g.alpha;
      }
      u32x2 o;
      o[0] = pack16x2<F16>(v[0], v[1]);
      o[1] = pack16x2<F16>(v[2], v[3]);
      if constexpr (STATS) {
        const float r4[4] = {unpack16_lo<F16>(o[0]), unpack16_hi<F16>(o[0]), unpack16_lo<F16>(o[1]), unpack16_hi<F16>(o[1])};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          sacc[j] += r4[j];
          sacc[4 + j] = fmaf(r4[j], r4[j], sacc[4 + j]);
        }
      }
      __builtin_amdgcn_raw_buffer_store_b64(o, ry, 4 * q + 4 <= g.cout ? (unsigned)(((oy * g.w + ox) * g.cout + 4 * q) * 2) : OOB, 0,
                                            TG_STORE_AUX);
    }
  }
  if constexpr (STATS) {
    // the 16 lanes of a group hold different pixels of the same 4 channels: butterfly over the pixel bits, then the four
    // waves through LDS in wave order; out[which][ch] as stats_flush writes it
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) sacc[i] += __shfl_xor(sacc[i], o, 64);
    }
    float* red = reinterpret_cast<float*>(sA);      // [wave][which][16]
    __syncthreads();
    if (c16 == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        red[(wid * 2 + 0) * 16 + 4 * q + j] = sacc[j];
        red[(wid * 2 + 1) * 16 + 4 * q + j] = sacc[4 + j];
      }
    }
    __syncthreads();
    if (tid < 32) {
      const int which = tid >> 4, ch = tid & 15;
      const float tsum = (red[(0 * 2 + which) * 16 + ch] + red[(1 * 2 + which) * 16 + ch]) +
                         (red[(2 * 2 + which) * 16 + ch] + red[(3 * 2 + which) * 16 + ch]);
      const int tpi = g.tiles_x * g.tiles_y;
      float* out = g.stats + ((size_t)(t_begin / tpi) * g.stat_chunks + (t_begin % tpi) / g.tiles_per_wg) * 2 * g.cout;
      if (ch < g.cout) out[(size_t)which * g.cout + ch] = tsum;
    }
  }
}

// The same wave program over concat(nearest_up2(x0), x1) with 32 + 32 channels (the generator's 256 x 256 concat conv,
// nets/pggan.py:69-76): both sources' halos are staged side by side (two LDS images, one barrier per tile), 18 A fragments
// (9 taps x 2 sources, 72 VGPRs) stay in registers; the sources are read in place as in the UPCAT tile kernels.
template <bool STATS, bool F16>
__global__ __launch_bounds__(256) void conv_thin16_upcat_kernel(const bf16* __restrict__ x0, const bf16* __restrict__ wp,
                                                                bf16* __restrict__ y, const TileGeom g) {
  constexpr int NT = 9, TW = 16, TH = 8, HWX = TW + 2, HH = TH + 2, KC = 32;
  constexpr int VPP = KC / 8, PS_A = THIN_PS;
  constexpr int AVEC = HH * HWX * VPP, ASLOTS = (AVEC + 255) / 256;
  constexpr int A_BYTES = (HH * HWX * PS_A + 15) & ~15;
  unsigned char* sA = tile_smem;      // [2 sources][A_BYTES]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int q = lane >> 4, c16 = lane & 15;
  const int c1 = g.cin - g.c0;      // 32 + 32 (checked by the launcher)

  const int wrow = NT * g.cin_pad;
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(wp, (unsigned)((size_t)g.cout * wrow * 2));
  bf16x8 wf[2][NT];
  int a_off[NT];
#pragma unroll
  for (int p = 0; p < NT; ++p) {
#pragma unroll
    for (int src = 0; src < 2; ++src)
      wf[src][p] = buf_load16(rw, c16 < g.cout ? (unsigned)((c16 * wrow + p * g.cin_pad + src * KC + q * 8) * 2) : OOB);
    a_off[p] = ((p / 3) * HWX + (p % 3)) * PS_A + q * 16;
  }
  const int a_base = ((wid * 2) * HWX + c16) * PS_A;

  int a_hy[ASLOTS], a_hx[ASLOTS], a_loff[ASLOTS];
#pragma unroll
  for (int s = 0; s < ASLOTS; ++s) {
    const int v = tid + s * 256;
    const int px = v / VPP, part = v % VPP;
    a_hy[s] = (v < AVEC) ? px / HWX : -100000;
    a_hx[s] = px % HWX;
    a_loff[s] = px * PS_A + part * 16;
  }
  int wg = blockIdx.x;
  const int nwg = gridDim.x;
  if ((nwg & 7) == 0) wg = (wg & 7) * (nwg >> 3) + (wg >> 3);
  const int t_begin = wg * g.tiles_per_wg;
  int t_end = t_begin + g.tiles_per_wg;
  if (t_end > g.nblk) t_end = g.nblk;
  const size_t img0_elems = (size_t)(g.h / 2) * (g.w / 2) * g.c0, img1_elems = (size_t)g.h * g.w * c1;
  const size_t out_img = (size_t)g.h * g.w * g.cout;

  struct Stage {
    bf16x8 ra[2][ASLOTS];
  };
  auto load_a = [&](Stage& st, int t) __attribute__((always_inline)) {
    const bool live = t < t_end;
    if (!live) t = t_begin;
    const int tx = t % g.tiles_x;
    const int r = t / g.tiles_x;
    const int ty = r % g.tiles_y;
    const int img = r / g.tiles_y;
    const int img1 = g.gsz ? (int)((g.perm >> (8 * (img / g.gsz))) & 0xffu) * g.gsz + img % g.gsz : img;
    const __amdgpu_buffer_rsrc_t r0 = make_rsrc(x0 + (size_t)img * img0_elems, (unsigned)(img0_elems * 2));
    const __amdgpu_buffer_rsrc_t r1 = make_rsrc(g.x1 + (size_t)img1 * img1_elems, (unsigned)(img1_elems * 2));
#pragma unroll
    for (int s = 0; s < ASLOTS; ++s) {
      const int iy = ty * TH + a_hy[s] - g.pad, ix = tx * TW + a_hx[s] - g.pad;
      const bool ok = live && iy >= 0 && iy < g.h && ix >= 0 && ix < g.w;
      const int part8 = ((tid + s * 256) % VPP) * 8;
      st.ra[0][s] = buf_load16(r0, ok ? (unsigned)((((iy >> 1) * (g.w >> 1) + (ix >> 1)) * g.c0 + part8) * 2) : OOB);
      st.ra[1][s] = buf_load16(r1, ok ? (unsigned)(((iy * g.w + ix) * c1 + part8) * 2) : OOB);
    }
  };

  float sacc[STATS ? 8 : 1];
  if constexpr (STATS) {
#pragma unroll
    for (int i = 0; i < 8; ++i) sacc[i] = 0.f;
  }
  bool first = true;
  Stage sa;
  load_a(sa, t_begin);
  for (int t = t_begin; t < t_end; ++t) {
    const int tx = t % g.tiles_x;
    const int r = t / g.tiles_x;
    const int ty = r % g.tiles_y;
    const int img = r / g.tiles_y;
    if (!first) __syncthreads();
    first = false;
#pragma unroll
    for (int src = 0; src < 2; ++src)
#pragma unroll
      for (int s = 0; s < ASLOTS; ++s)
        if (s < ASLOTS - 1 || tid + s * 256 < AVEC) *reinterpret_cast<bf16x8*>(sA + src * A_BYTES + a_loff[s]) = sa.ra[src][s];
    __syncthreads();
    load_a(sa, t + 1);
    f32x4 acc[2];
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) acc[pb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int src = 0; src < 2; ++src)
#pragma unroll
      for (int p = 0; p < NT; ++p)
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
          const bf16x8 xf = *reinterpret_cast<const bf16x8*>(sA + src * A_BYTES + a_base + pb * HWX * PS_A + a_off[p]);
          acc[pb] = mfma_16x16x32<F16>(wf[src][p], xf, acc[pb]);
        }
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(y + (size_t)img * out_img, (unsigned)(out_img * 2));
    const int ox = tx * TW + c16;
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
      const int oy = ty * TH + wid * 2 + pb;
      u32x2 o;
      o[0] = pack16x2<F16>(acc[pb][0], acc[pb][1]);
      o[1] = pack16x2<F16>(acc[pb][2], acc[pb][3]);
      if constexpr (STATS) {
        const float r4[4] = {unpack16_lo<F16>(o[0]), unpack16_hi<F16>(o[0]), unpack16_lo<F16>(o[1]), unpack16_hi<F16>(o[1])};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          sacc[j] += r4[j];
          sacc[4 + j] = fmaf(r4[j], r4[j], sacc[4 + j]);
        }
      }
      __builtin_amdgcn_raw_buffer_store_b64(o, ry, 4 * q + 4 <= g.cout ? (unsigned)(((oy * g.w + ox) * g.cout + 4 * q) * 2) : OOB, 0,
                                            TG_STORE_AUX);
    }
  }
  if constexpr (STATS) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) sacc[i] += __shfl_xor(sacc[i], o, 64);
    }
    float* red = reinterpret_cast<float*>(sA);
    __syncthreads();
    if (c16 == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        red[(wid * 2 + 0) * 16 + 4 * q + j] = sacc[j];
        red[(wid * 2 + 1) * 16 + 4 * q + j] = sacc[4 + j];
      }
    }
    __syncthreads();
    if (tid < 32) {
      const int which = tid >> 4, ch = tid & 15;
      const float tsum = (red[(0 * 2 + which) * 16 + ch] + red[(1 * 2 + which) * 16 + ch]) +
                         (red[(2 * 2 + which) * 16 + ch] + red[(3 * 2 + which) * 16 + ch]);
      const int tpi = g.tiles_x * g.tiles_y;
      float* out = g.stats + ((size_t)(t_begin / tpi) * g.stat_chunks + (t_begin % tpi) / g.tiles_per_wg) * 2 * g.cout;
      if (ch < g.cout) out[(size_t)which * g.cout + ch] = tsum;
    }
  }
}

// 3x3 layers with <= 16 output channels and one 16- / 32-channel chunk (or the 32 + 32 concat) go to the thin-output kernels;
// TG_THIN16=0: the 32-wide-block kernels instead (A/B switch, read at every call: two captures in one process can differ)
inline bool thin16_on() { return tg_tune("TG_THIN16", 1) != 0; }
inline bool thin16_takes(const TileGeom& g) {
  return thin16_on() && g.cout <= 16 && g.cout % 4 == 0 && (g.cin_pad == 16 || g.cin_pad == 32) && g.cin == g.cin_pad && !g.ypool &&
         !g.up_src && !g.up_out && !g.skip_out && !(g.mask && (g.epilogue & TG_EPI_BIAS));
}

template <int KC>
int launch_thin16(const TileGeom& g0, const bf16* x, const bf16* wp, const float* bias, bf16* y, hipStream_t s) {
  TileGeom g = g0;
  g.tiles_x = g.w / 16;
  g.tiles_y = g.h / 8;
  g.nblk = g.tiles_x * g.tiles_y * g.n;
  int tpw = g.nblk / (256 * 4);
  if (tpw < 1) tpw = 1;
  if (tpw > 16) tpw = 16;
  const bool stats = g.stats || g.chunks_query;
  if (stats) {
    const int tpi = g.tiles_x * g.tiles_y;
    while (tpi % tpw) --tpw;
    if (g.chunks_query) {
      *g.chunks_query = tpi / tpw;
      return TG_OK;
    }
    TG_CHECK(g.stat_chunks == tpi / tpw, TG_EINVAL, "conv_thin16: stat_chunks %d, this dispatch writes %d", g.stat_chunks, tpi / tpw);
    TG_CHECK(g.epilogue == 0 && !g.mask, TG_ENOSUP, "conv_thin16: statistics come with the plain epilogue only");
  }
  g.tiles_per_wg = tpw;
  const int nwg = (g.nblk + tpw - 1) / tpw;
  const size_t lds = (size_t)((10 * 18 * THIN_PS + 15) & ~15);
  tg_note_kernel(g.f16 ? "conv_thin16_kernel<%d%s,f16>" : "conv_thin16_kernel<%d%s>", KC, stats ? ",stats" : "");
#define TG_THIN_LAUNCH(MODE_, EPI_)                                                                                              \
  do {                                                                                                                           \
    if (g.f16) hipLaunchKernelGGL((conv_thin16_kernel<KC, MODE_, true, EPI_>), dim3(nwg), dim3(256), lds, s, x, wp, bias, y, g); \
    else hipLaunchKernelGGL((conv_thin16_kernel<KC, MODE_, false, EPI_>), dim3(nwg), dim3(256), lds, s, x, wp, bias, y, g);     \
  } while (0)
  if (stats) TG_THIN_LAUNCH(1, 0);
  else if (g.mask) TG_THIN_LAUNCH(0, 2);
  else if (g.epilogue & TG_EPI_BIAS) TG_THIN_LAUNCH(0, 1);
  else TG_THIN_LAUNCH(0, 0);
#undef TG_THIN_LAUNCH
  TG_LAUNCH_CHECK("conv_thin16");
  return TG_OK;
}

int launch_thin16_upcat(const TileGeom& g0, const bf16* x0, const bf16* wp, bf16* y, hipStream_t s) {
  TileGeom g = g0;
  g.tiles_x = g.w / 16;
  g.tiles_y = g.h / 8;
  g.nblk = g.tiles_x * g.tiles_y * g.n;
  int tpw = g.nblk / (256 * 4);
  if (tpw < 1) tpw = 1;
  if (tpw > 16) tpw = 16;
  const bool stats = g.stats || g.chunks_query;
  if (stats) {
    const int tpi = g.tiles_x * g.tiles_y;
    while (tpi % tpw) --tpw;
    if (g.chunks_query) {
      *g.chunks_query = tpi / tpw;
      return TG_OK;
    }
    TG_CHECK(g.stat_chunks == tpi / tpw, TG_EINVAL, "conv_thin16(upcat): stat_chunks %d, this dispatch writes %d", g.stat_chunks,
             tpi / tpw);
  }
  g.tiles_per_wg = tpw;
  const int nwg = (g.nblk + tpw - 1) / tpw;
  const size_t lds = 2 * (size_t)((10 * 18 * THIN_PS + 15) & ~15);
  tg_note_kernel(g.f16 ? "conv_thin16_upcat_kernel<%s,f16>" : "conv_thin16_upcat_kernel<%s>", stats ? "stats" : "plain");
  if (stats) {
    if (g.f16) hipLaunchKernelGGL((conv_thin16_upcat_kernel<true, true>), dim3(nwg), dim3(256), lds, s, x0, wp, y, g);
    else hipLaunchKernelGGL((conv_thin16_upcat_kernel<true, false>), dim3(nwg), dim3(256), lds, s, x0, wp, y, g);
  } else {
    if (g.f16) hipLaunchKernelGGL((conv_thin16_upcat_kernel<false, true>), dim3(nwg), dim3(256), lds, s, x0, wp, y, g);
    else hipLaunchKernelGGL((conv_thin16_upcat_kernel<false, false>), dim3(nwg), dim3(256), lds, s, x0, wp, y, g);
  }
  TG_LAUNCH_CHECK("conv_thin16(upcat)");
  return TG_OK;
}

template <int KH, int KC, int BN, int NCH>
int launch_tile_wres(const TileGeom& g0, const bf16* x, const bf16* wp, const float* bias, bf16* y, hipStream_t s) {
  TileGeom g = g0;
  constexpr int HWX = 16 + KH - 1, HH = 8 + KH - 1;
  g.tiles_x = g.w / 16;
  g.tiles_y = g.h / 8;
  g.nblk = g.tiles_x * g.tiles_y * g.n;
  // UPBOTH (kernel MODE 4): a 32 + 32 concat in ONE 64-channel block that walks the tiles of the n1 skip images
  const bool upboth = BN == 64 && KH == 3 && (g.up_out || g.skip_out) && g.c0 == 32 && g.cout == 64 && g.n1 > 0;
  if (upboth) g.nblk = g.tiles_x * g.tiles_y * g.n1;
  const int ny = (g.cout + BN - 1) / BN;
  int tpw = g.nblk * ny / (256 * 4);          // aim for ~4 workgroups per CU over the whole grid
  if (tpw < 1) tpw = 1;
  if (tpw > 16) tpw = 16;
  const bool stats = g.stats || g.chunks_query;
  if (stats) {      // a workgroup must stay inside one image
    const int tpi = g.tiles_x * g.tiles_y;
    while (tpi % tpw) --tpw;
    if (g.chunks_query) {
      *g.chunks_query = (KH == 3) ? tpi / tpw : 0;
      return TG_OK;
    }
    TG_CHECK(g.stat_chunks == tpi / tpw, TG_EINVAL, "conv_tile(wres): stat_chunks %d, this dispatch writes %d", g.stat_chunks,
             tpi / tpw);
  }
  g.tiles_per_wg = tpw;
  const int nwg = (g.nblk + tpw - 1) / tpw;
  const size_t lds = (size_t)((HH * HaloLds<KC, HWX>::ROW + 15) & ~15) + (size_t)NCH * BN * (KH * KH * KC * 2 + 16);
  TG_CHECK(lds <= 64 * 1024, TG_ENOSUP, "conv_tile(wres): LDS %zu too large", lds);
  TG_CHECK(!(g.mask && (g.epilogue & TG_EPI_BIAS)), TG_ENOSUP, "conv_tile(wres): a bias and a mask epilogue do not come together");
  const char* fmt = g.f16 ? ",f16" : "";
// the epilogue variant MODE_ in the element format of the call (both formats are instantiated for every epilogue)
#define TG_WRES_LAUNCH_E(MODE_, EPI_)                                                                                           \
  do {                                                                                                                          \
    if (g.f16)                                                                                                                  \
      hipLaunchKernelGGL((conv_tile_wres_kernel<KH, KC, BN, NCH, MODE_, true, EPI_>), dim3(nwg, ny), dim3(256), lds, s, x, wp, \
                         bias, y, g);                                                                                           \
    else                                                                                                                        \
      hipLaunchKernelGGL((conv_tile_wres_kernel<KH, KC, BN, NCH, MODE_, false, EPI_>), dim3(nwg, ny), dim3(256), lds, s, x,    \
                         wp, bias, y, g);                                                                                       \
  } while (0)
// the variant that carries only the epilogue operands this call has (bias and mask never come together)
#define TG_WRES_LAUNCH(MODE_)                                     \
  do {                                                            \
    if (g.mask) TG_WRES_LAUNCH_E(MODE_, 2);                       \
    else if (g.epilogue & TG_EPI_BIAS) TG_WRES_LAUNCH_E(MODE_, 1); \
    else TG_WRES_LAUNCH_E(MODE_, 0);                              \
  } while (0)
  if (g.up_src) {
    if constexpr (KH == 3 && KC == 32 && NCH == 1) {
      TG_CHECK(g.epilogue == 0 && !g.ypool && !stats && !g.up_out && !g.skip_out && g.cin % 32 == 0, TG_ENOSUP,
               "conv_tile(wres): the unpooling input comes with the plain / masked epilogue and 32-channel chunks only");
      tg_note_kernel(g.up_z ? "conv_tile_wres_kernel<%d,%d,%d,%d,unpoolz%s>" : "conv_tile_wres_kernel<%d,%d,%d,%d,unpool%s>", KH, KC, BN,
                     NCH, fmt);
      if (g.up_z) {
        if (g.mask) TG_WRES_LAUNCH_E(6, 2);
        else TG_WRES_LAUNCH_E(6, 0);
      } else if (g.mask) TG_WRES_LAUNCH_E(5, 2);
      else TG_WRES_LAUNCH_E(5, 0);
    } else {
      TG_CHECK(false, TG_ENOSUP, "conv_tile(wres): the unpooling input is built for 3x3, 32-channel chunks only");
    }
  } else if (g.up_out || g.skip_out) {
    if constexpr (KH == 3) {
      TG_CHECK(g.epilogue == 0 && !g.mask && !g.ypool && !stats, TG_ENOSUP, "conv_tile(wres): the concat backward comes with the plain epilogue only");
      if constexpr (BN == 64) {
        if (upboth) {
          tg_note_kernel("conv_tile_wres_kernel<%d,%d,%d,%d,upboth%s>", KH, KC, BN, NCH, fmt);
          TG_WRES_LAUNCH_E(4, 0);
          TG_LAUNCH_CHECK("conv_tile_wres");
          return TG_OK;
        }
      }
      tg_note_kernel("conv_tile_wres_kernel<%d,%d,%d,%d,upbwd%s>", KH, KC, BN, NCH, fmt);
      TG_WRES_LAUNCH_E(3, 0);
    } else {
      TG_CHECK(false, TG_ENOSUP, "conv_tile(wres): the concat backward is built for 3x3 only");
    }
  } else if (stats) {
    if constexpr (KH == 3) {
      TG_CHECK(g.epilogue == 0 && !g.mask && !g.ypool, TG_ENOSUP, "conv_tile(wres): statistics come with the plain epilogue only");
      tg_note_kernel("conv_tile_wres_kernel<%d,%d,%d,%d,stats%s>", KH, KC, BN, NCH, fmt);
      TG_WRES_LAUNCH_E(1, 0);
    } else {
      TG_CHECK(false, TG_ENOSUP, "conv_tile(wres): statistics epilogue is built for 3x3 only");
    }
  } else if (g.ypool) {
    if constexpr (KH == 3) {
      TG_CHECK(!g.mask, TG_ENOSUP, "conv_tile(wres): the pooled output is a forward feature");
      tg_note_kernel("conv_tile_wres_kernel<%d,%d,%d,%d,pool%s>", KH, KC, BN, NCH, fmt);
      if (g.epilogue & TG_EPI_BIAS) TG_WRES_LAUNCH_E(2, 1);
      else TG_WRES_LAUNCH_E(2, 0);
    } else {
      TG_CHECK(false, TG_ENOSUP, "conv_tile(wres): pooled output is built for 3x3 only");
    }
  } else {
    tg_note_kernel(g.f16 ? "conv_tile_wres_kernel<%d,%d,%d,%d,f16>" : "conv_tile_wres_kernel<%d,%d,%d,%d>", KH, KC, BN, NCH);
    TG_WRES_LAUNCH(0);
  }
#undef TG_WRES_LAUNCH
#undef TG_WRES_LAUNCH_E
  TG_LAUNCH_CHECK("conv_tile_wres");
  return TG_OK;
}

// one instantiation of the tile kernel: raises its dynamic-LDS limit once, notes its name, launches
template <int KH, int KC, int BN, int MT, bool UPCAT, int MODE, bool F16>
int launch_tile_variant(const TileGeom& g, size_t lds, const bf16* x, const bf16* wp, const float* bias, bf16* y, hipStream_t s) {
  auto kern = conv_tile_kernel<KH, KC, BN, MT, UPCAT, MODE, F16>;
  if (lds > 64 * 1024) {
    static unsigned long long raised = 0;      // per instantiation, one bit per device
    if (tg_first_on_device(&raised)) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
          hipSuccess) {
        tg_set_error("conv_tile: cannot raise dynamic LDS to %zu", lds);
        return TG_ELAUNCH;
      }
    }
  }
  // names as before for the bf16 kernels (tests/golden/bench_dispatch_kernels.json); ",f16" marks the half instantiation
  static const char* const mode_tag[6] = {"", ",stats", ",pool", ",upbwd", ",unpool", ",unpoolz"};
  // ",sets": one launch over several weight sets (TileGeom::npg)
  if (F16 && MODE == 0 && !UPCAT) tg_note_kernel("conv_tile_kernel<%d,%d,%d,%d,f16%s>", KH, KC, BN, MT, g.npg ? ",sets" : "");
  else tg_note_kernel("conv_tile_kernel<%d,%d,%d,%d%s%s%s%s>", KH, KC, BN, MT, UPCAT ? ",upcat" : "", mode_tag[MODE], F16 ? ",f16" : "",
                      g.npg ? ",sets" : "");
  hipLaunchKernelGGL(kern, dim3(g.nblk, (g.cout + BN - 1) / BN), dim3(256), lds, s, x, wp, bias, y, g);
  TG_LAUNCH_CHECK("conv_tile");
  return TG_OK;
}

template <int KH, int KC, int BN, int MT, bool UPCAT = false>
int launch_tile(const TileGeom& g0, const bf16* x, const bf16* wp, const float* bias, bf16* y, hipStream_t s) {
  TileGeom g = g0;
  constexpr int TH = 8 * MT, HWX = 16 + KH - 1, HH = TH + KH - 1;
  g.tiles_x = g.w / 16;
  g.tiles_y = g.h / TH;
  g.nblk = g.tiles_x * g.tiles_y * g.n;
  const size_t lds = (size_t)((HH * HaloLds<KC, HWX>::ROW + 15) & ~15) + (size_t)BN * (KH * KH * KC * 2 + 16);
  TG_CHECK(lds <= 160 * 1024, TG_ENOSUP, "conv_tile: LDS %zu too large", lds);
  if (g.chunks_query) {
    *g.chunks_query = (KH == 3) ? g.tiles_x * g.tiles_y : 0;
    return TG_OK;
  }
  if (g.up_src) {
    if constexpr (KH == 3 && !UPCAT && KC == 32) {
      TG_CHECK(g.epilogue == 0 && !g.stats && !g.ypool && !g.up_out && !g.skip_out && g.cin % 32 == 0, TG_ENOSUP,
               "conv_tile: the unpooling input comes with the plain / masked epilogue and 32-channel chunks only");
      if (g.up_z)
        return g.f16 ? launch_tile_variant<KH, KC, BN, MT, false, 5, true>(g, lds, x, wp, bias, y, s)
                     : launch_tile_variant<KH, KC, BN, MT, false, 5, false>(g, lds, x, wp, bias, y, s);
      return g.f16 ? launch_tile_variant<KH, KC, BN, MT, false, 4, true>(g, lds, x, wp, bias, y, s)
                   : launch_tile_variant<KH, KC, BN, MT, false, 4, false>(g, lds, x, wp, bias, y, s);
    } else {
      TG_CHECK(false, TG_ENOSUP, "conv_tile: the unpooling input is built for plain 3x3 backward-data with 32-channel chunks only");
    }
  }
  if (g.up_out || g.skip_out) {
    if constexpr (KH == 3 && !UPCAT) {
      TG_CHECK(g.epilogue == 0 && !g.mask && !g.stats && !g.ypool, TG_ENOSUP, "conv_tile: the concat backward comes with the plain epilogue only");
      return g.f16 ? launch_tile_variant<KH, KC, BN, MT, false, 3, true>(g, lds, x, wp, bias, y, s)
                   : launch_tile_variant<KH, KC, BN, MT, false, 3, false>(g, lds, x, wp, bias, y, s);
    } else {
      TG_CHECK(false, TG_ENOSUP, "conv_tile: the concat backward is built for plain 3x3 backward-data only");
    }
  }
  if (g.stats) {
    if constexpr (KH == 3) {
      TG_CHECK(g.epilogue == 0 && !g.mask, TG_ENOSUP, "conv_tile: statistics come with the plain epilogue only");
      TG_CHECK(g.stat_chunks == g.tiles_x * g.tiles_y, TG_EINVAL, "conv_tile: stat_chunks %d, this dispatch writes %d",
               g.stat_chunks, g.tiles_x * g.tiles_y);
      return g.f16 ? launch_tile_variant<KH, KC, BN, MT, UPCAT, 1, true>(g, lds, x, wp, bias, y, s)
                   : launch_tile_variant<KH, KC, BN, MT, UPCAT, 1, false>(g, lds, x, wp, bias, y, s);
    } else {
      TG_CHECK(false, TG_ENOSUP, "conv_tile: statistics epilogue is built for 3x3 only");
    }
  }
  if (g.ypool) {
    if constexpr (KH == 3 && !UPCAT) {
      TG_CHECK(!g.mask, TG_ENOSUP, "conv_tile: the pooled output is a forward feature");
      return g.f16 ? launch_tile_variant<KH, KC, BN, MT, false, 2, true>(g, lds, x, wp, bias, y, s)
                   : launch_tile_variant<KH, KC, BN, MT, false, 2, false>(g, lds, x, wp, bias, y, s);
    } else {
      TG_CHECK(false, TG_ENOSUP, "conv_tile: pooled output is built for plain 3x3 convs only");
    }
  }
  return g.f16 ? launch_tile_variant<KH, KC, BN, MT, UPCAT, 0, true>(g, lds, x, wp, bias, y, s)
               : launch_tile_variant<KH, KC, BN, MT, UPCAT, 0, false>(g, lds, x, wp, bias, y, s);
}

// forward over concat(nearest_up2(x), x1): 3x3, both channel counts multiples of 32
int dispatch_tile_upcat(const TileGeom& g, const bf16* x, const bf16* wp, const float* bias, bf16* y, hipStream_t s) {
  if (thin16_on() && !bias && g.cout <= 16 && g.cout % 4 == 0 && g.c0 == 32 && g.cin - g.c0 == 32 && g.cin_pad == 64 &&
      (g.w / 16) * (g.h / 8) * g.n >= 2048)
    return launch_thin16_upcat(g, x, wp, y, s);
  const bool wide = g.cout > 32;
  const int tiles1 = (g.w / 16) * (g.h / 8) * g.n * ((g.cout + (wide ? 63 : 31)) / (wide ? 64 : 32));
  const bool mt2 = (g.h % 16 == 0) && tiles1 >= 2 * 2 * 256;
  if (wide) return mt2 ? launch_tile<3, 32, 64, 2, true>(g, x, wp, bias, y, s) : launch_tile<3, 32, 64, 1, true>(g, x, wp, bias, y, s);
  return mt2 ? launch_tile<3, 32, 32, 2, true>(g, x, wp, bias, y, s) : launch_tile<3, 32, 32, 1, true>(g, x, wp, bias, y, s);
}

// backward-data of the UPCAT forward with the upsample / concat adjoint in the epilogue (TileGeom::up_out): dispatch_tile's
// rules, with 64-channel output blocks only where both halves of the concat are multiples of 64 (a block is all "up" or
// all "skip")
int dispatch_tile_upbwd(const TileGeom& g, const bf16* gy, const bf16* wp, hipStream_t s) {
  const int c1 = g.cout - g.c0;
  const bool wide = g.c0 % 64 == 0 && c1 % 64 == 0 && (g.w / 16) * (g.h / 8) * g.n * ((g.cout + 63) / 64) >= 1024;
  const int tiles1 = (g.w / 16) * (g.h / 8) * g.n * ((g.cout + (wide ? 63 : 31)) / (wide ? 64 : 32));
  const bool mt2 = (g.h % 16 == 0) && tiles1 >= 2 * 2 * 256 && g.cin_pad >= 64;
  if (tiles1 >= 2048) {
    // a 32 + 32 concat: both halves in one 64-channel block, every gy tile staged once (launch_tile_wres picks MODE 4)
    const bool both = g.c0 == 32 && c1 == 32 && g.n1 > 0;
    if (g.cin_pad == 16 && both) return launch_tile_wres<3, 16, 64, 1>(g, gy, wp, nullptr, nullptr, s);
    if (g.cin_pad == 16) return wide ? launch_tile_wres<3, 16, 64, 1>(g, gy, wp, nullptr, nullptr, s) : launch_tile_wres<3, 16, 32, 1>(g, gy, wp, nullptr, nullptr, s);
    if (g.cin_pad == 32) return wide ? launch_tile_wres<3, 32, 64, 1>(g, gy, wp, nullptr, nullptr, s) : launch_tile_wres<3, 32, 32, 1>(g, gy, wp, nullptr, nullptr, s);
  }
  if (g.cin_pad % 32 == 0) {
    if (wide) return mt2 ? launch_tile<3, 32, 64, 2>(g, gy, wp, nullptr, nullptr, s) : launch_tile<3, 32, 64, 1>(g, gy, wp, nullptr, nullptr, s);
    return mt2 ? launch_tile<3, 32, 32, 2>(g, gy, wp, nullptr, nullptr, s) : launch_tile<3, 32, 32, 1>(g, gy, wp, nullptr, nullptr, s);
  }
  if (wide) return launch_tile<3, 16, 64, 1>(g, gy, wp, nullptr, nullptr, s);
  return launch_tile<3, 16, 32, 1>(g, gy, wp, nullptr, nullptr, s);
}

// does dispatch_tile send this geometry to a kernel whose workgroups hold ONE weight slice over several tiles / images
// (conv_tile_wres, conv_thin16)?  Those do not select a weight set per image.
static bool tile_leaves_plain_kernel(const TileGeom& g) {
  const bool wide = g.cout > 32 && (g.w / 16) * (g.h / 8) * g.n * ((g.cout + 63) / 64) >= 1024;
  const int tiles1 = (g.w / 16) * (g.h / 8) * g.n * ((g.cout + (wide ? 63 : 31)) / (wide ? 64 : 32));
  return tiles1 >= 2048 && (g.cin_pad == 16 || g.cin_pad == 32);
}

template <int KH>
int dispatch_tile(const TileGeom& g, const bf16* x, const bf16* wp, const float* bias, bf16* y, hipStream_t s) {
  // 64-channel blocks halve the pixel staging per output, but a grid under ~4 workgroups per CU wants 32-channel
  // blocks (kbench 16x16x256 n16: 15.7 -> 10.6 us; 32x32x128 n48: 21.9 -> 20.2 us; above 1024 blocks 64 wins)
  const bool wide = g.cout > 32 && (g.w / 16) * (g.h / 8) * g.n * ((g.cout + 63) / 64) >= 1024;
  const int tiles1 = (g.w / 16) * (g.h / 8) * g.n * ((g.cout + (wide ? 63 : 31)) / (wide ? 64 : 32));
  // two sub-tiles per wave (256-pixel workgroup tile) halve the weight staging per pixel; use them
  // when that still leaves >= 2 workgroups per CU and the map is tall enough
  const bool mt2 = (g.h % 16 == 0) && tiles1 >= 2 * 2 * 256 && g.cin_pad >= 64;
  // thin layers with many tiles: weights resident in LDS, several tiles per workgroup
  // weight-set groups: the plain tile kernel only (tg_conv_tile_grouped_native answers for this dispatch)
  TG_CHECK(!g.npg || !tile_leaves_plain_kernel(g), TG_ENOSUP, "conv_tile: weight-set groups on a weight-resident / thin-output dispatch");
  if constexpr (KH == 3) {
    if (tiles1 >= 2048 && thin16_takes(g)) return g.cin_pad == 16 ? launch_thin16<16>(g, x, wp, bias, y, s) : launch_thin16<32>(g, x, wp, bias, y, s);
  }
  if (tiles1 >= 2048 && !(g.up_src && g.cin_pad != 32)) {
    if (g.cin_pad == 16) return wide ? launch_tile_wres<KH, 16, 64, 1>(g, x, wp, bias, y, s) : launch_tile_wres<KH, 16, 32, 1>(g, x, wp, bias, y, s);
    if (g.cin_pad == 32) return wide ? launch_tile_wres<KH, 32, 64, 1>(g, x, wp, bias, y, s) : launch_tile_wres<KH, 32, 32, 1>(g, x, wp, bias, y, s);
  }
  if (g.cin_pad % 32 == 0) {
    if (wide) return mt2 ? launch_tile<KH, 32, 64, 2>(g, x, wp, bias, y, s) : launch_tile<KH, 32, 64, 1>(g, x, wp, bias, y, s);
    return mt2 ? launch_tile<KH, 32, 32, 2>(g, x, wp, bias, y, s) : launch_tile<KH, 32, 32, 1>(g, x, wp, bias, y, s);
  }
  if (wide) return launch_tile<KH, 16, 64, 1>(g, x, wp, bias, y, s);
  return launch_tile<KH, 16, 32, 1>(g, x, wp, bias, y, s);
}

}  // namespace

// Shapes this kernel takes: square-kernel 1x1 / 3x3, stride 1, SAME, h % 8 == 0, w % 16 == 0.
bool tg_conv_tile_supported(int h, int w, int hout, int wout, int kh, int kw, int pad_t, int pad_l) {
  if (kh != kw || (kh != 1 && kh != 3)) return false;
  if (h != hout || w != wout) return false;
  if (pad_t != pad_l) return false;
  return (h % 8 == 0) && (w % 16 == 0);
}

// Does a grouped call of this shape (n = the whole batch) stay on conv_tile_kernel, which picks the weight set per image?
bool tg_conv_tile_grouped_native(int n, int h, int w, int cin, int cout) {
  TileGeom g;
  g.n = n; g.h = h; g.w = w; g.cin = cin; g.cout = cout;
  g.cin_pad = (cin + 15) / 16 * 16;
  return !tile_leaves_plain_kernel(g);
}

int tg_conv_tile_run(int n, int h, int w, int cin, int cout, int k, int pad, int epilogue, float alpha, const void* x,
                     const void* wp, const float* bias, void* y, hipStream_t s, const void* mask, float* stats,
                     int stat_chunks, int* chunks_query, void* ypool, void* ymask, const void* up_src, const void* up_signs,
                     float up_alpha, void* up_store, const void* up_z, int groups, size_t wset_elems) {
  TileGeom g;
  g.n = n; g.h = h; g.w = w; g.cin = cin; g.cout = cout;
  g.cin_pad = (cin + 15) / 16 * 16;
  g.pad = pad;
  g.npg = groups > 1 ? n / groups : 0;
  g.wgs = groups > 1 ? (unsigned)wset_elems : 0u;
  g.tiles_x = g.tiles_y = g.nblk = 0;
  g.epilogue = epilogue;
  g.alpha = alpha;
  g.x1 = nullptr;
  g.c0 = g.gsz = 0;
  g.perm = 0;
  g.mask = (const bf16*)mask;
  g.stats = stats;
  g.stat_chunks = stat_chunks;
  g.chunks_query = chunks_query;
  g.ypool = (bf16*)ypool;
  g.ymask = (unsigned char*)ymask;
  g.up_out = g.skip_out = nullptr;
  g.n1 = 0;
  g.up_src = (const bf16*)up_src;
  g.up_signs = (const unsigned char*)up_signs;
  g.up_z = (const bf16*)up_z;
  g.up_alpha = up_alpha;
  g.up_store = (bf16*)up_store;
  g.f16 = tg_elem_f16();      // the descriptor's dtype, noted by the C-ABI entry point
  if (k == 1) return dispatch_tile<1>(g, (const bf16*)x, (const bf16*)wp, bias, (bf16*)y, s);
  return dispatch_tile<3>(g, (const bf16*)x, (const bf16*)wp, bias, (bf16*)y, s);
}

// y = conv3x3_same(concat(nearest_up2(x0 [n,h/2,w/2,c0]), x1 [n1,h,w,c1])) without materialising the concat.
bool tg_conv_tile_upcat_supported(int h, int w, int c0, int c1, int cout) {
  return (h % 8 == 0) && (w % 16 == 0) && c0 > 0 && c1 > 0 && c0 % 32 == 0 && c1 % 32 == 0 && cout % 8 == 0;
}

int tg_conv_tile_upcat_run(int n, int h, int w, int c0, int c1, int cout, int gsz, unsigned perm, const void* x0,
                           const void* x1, const void* wp, void* y, hipStream_t s, float* stats, int stat_chunks,
                           int* chunks_query) {      // element format: tg_elem_f16(), set by the C-ABI entry point
  TileGeom g;
  g.n = n; g.h = h; g.w = w; g.cin = c0 + c1; g.cout = cout;
  g.cin_pad = g.cin;
  g.pad = 1;
  g.npg = 0;
  g.wgs = 0;
  g.tiles_x = g.tiles_y = g.nblk = 0;
  g.tiles_per_wg = 0;
  g.epilogue = 0;
  g.alpha = 0.f;
  g.x1 = (const bf16*)x1;
  g.c0 = c0;
  g.gsz = gsz;
  g.perm = perm;
  g.mask = nullptr;
  g.stats = stats;
  g.stat_chunks = stat_chunks;
  g.chunks_query = chunks_query;
  g.ypool = nullptr;
  g.ymask = nullptr;
  g.up_out = g.skip_out = nullptr;
  g.n1 = 0;
  g.up_src = nullptr;
  g.up_signs = nullptr;
  g.up_z = nullptr;
  g.up_alpha = 0.f;
  g.up_store = nullptr;
  g.f16 = tg_elem_f16();
  return dispatch_tile_upcat(g, (const bf16*)x0, (const bf16*)wp, nullptr, (bf16*)y, s);
}

// (g0 [n,h/2,w/2,c0], g1 [n1,h,w,c1]) = the adjoint of concat(nearest_up2(.), skip) applied to conv3x3^T(gy [n,h,w,cout], w):
// the input gradient of tg_conv_tile_upcat_run's conv.  wp: the backward-data pack (mode 1) of the conv's kernel.
int tg_conv_tile_upcat_bwd_run(int n, int h, int w, int c0, int c1, int cout, int gsz, unsigned perm, int n1, const void* gy,
                               const void* wp, void* g0, void* g1, hipStream_t s) {
  TileGeom g;
  g.n = n; g.h = h; g.w = w; g.cin = cout; g.cout = c0 + c1;      // a conv over gy: cout -> c0 + c1 channels
  g.cin_pad = (cout + 15) / 16 * 16;
  g.pad = 1;
  g.npg = 0;
  g.wgs = 0;
  g.tiles_x = g.tiles_y = g.nblk = 0;
  g.tiles_per_wg = 0;
  g.epilogue = 0;
  g.alpha = 1.f;
  g.x1 = nullptr;
  g.c0 = c0;
  g.gsz = gsz;
  g.perm = perm;
  g.mask = nullptr;
  g.stats = nullptr;
  g.stat_chunks = 0;
  g.chunks_query = nullptr;
  g.ypool = nullptr;
  g.ymask = nullptr;
  g.up_out = (bf16*)g0;
  g.skip_out = (bf16*)g1;
  g.n1 = n1;
  g.up_src = nullptr;
  g.up_signs = nullptr;
  g.up_z = nullptr;
  g.up_alpha = 0.f;
  g.up_store = nullptr;
  g.f16 = tg_elem_f16();
  return dispatch_tile_upbwd(g, (const bf16*)gy, (const bf16*)wp, s);
}
