// SAGAN self-attention (libs/self_attention.py:24-70: s = f g^T over the h*w positions, beta = softmax(s), o = beta h)
// without the [N x N] map: flash-style forward, first-order backward and second-order backward (the gradient penalty's)
// on the MFMA units.
//
// Shapes of the layer: N = h*w positions (4096 at 64x64), d_qk = c/8 (8 or 16), d_v = c (64..256).  All three kernels
// put the SOFTMAX ROWS' OWNERS ON THE LANES so that no reduction crosses lanes:
//   forward / dQ kernel: a wave owns 32 queries; S^T = K Q^T is one 32x32x16 MFMA per 32-key block (A = 32 keys x
//     d_qk padded to 16, B = d_qk x 32 queries), so lane l31 holds the scores of ITS query against 16 keys and its
//     partner lane l31 + 32 the other 16: the running max / sum are lane-local plus one half-wave exchange;
//   dK / dV kernel: a wave owns 32 keys, S = Q K^T tile per 32-query block, the per-query quantities (log-sum-exp, D)
//     are per accumulator ROW there and are read as broadcast loads.
// P (or dS) goes back into the next MFMA as the B operand after packing to 16 bit and the v_permlane32_swap of the conv
// epilogues: afterwards a lane of half kgrp holds 8 + 8 consecutive reduction indices (16 kgrp + 0..7 and + 8..15).
//
// Operand layout.  An MFMA A operand puts matrix ROWS on the lanes, so fetching it from a row-major tensor touches 32
// (or, for a transposed operand, 64) different cache lines per 1-KB wave load and the texture path, not the MFMA pipe,
// sets the pace (first version: 12 % MFMA utilisation).  The per-tile operands (V / dO as "rows x features" fragments,
// V^T / dO^T as "features x positions" fragments) are therefore re-packed once per call into FRAGMENT ORDER -- the 64
// lanes' 16-byte pieces of one fragment contiguous -- by two small kernels into a caller-provided workspace; every
// in-loop load is then one fully coalesced 1-KB read.  Q / K rows are 16 (32) bytes, already dense.  No LDS: the four
// waves of a workgroup read the same fragments through L1.
//
// MFMA layouts used (as in conv_tile.hip): A lane = row l%32, k = 8 (l/32) + i; B lane = column l%32, k = 8 (l/32) + i;
// D lane = column l%32, register r = row 8 (r/4) + 4 (l/32) + r%4.
#include "tg_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

struct FlashGeom {
  int n, len, dk, dv;      // images, positions, d_qk (8 | 16), d_v (multiple of 32)
};

__device__ __forceinline__ bf16x8 zero_frag() {
  u32x4 z;
  z[0] = z[1] = z[2] = z[3] = 0u;
  return __builtin_bit_cast(bf16x8, z);
}
__device__ __forceinline__ bf16x8 load_frag(const bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }
template <bool F16>
__device__ __forceinline__ bf16x8 ones_frag() {
  u32x4 z;
  z[0] = z[1] = z[2] = z[3] = ones16x2<F16>();
  return __builtin_bit_cast(bf16x8, z);
}
// exp(x - m) as ONE fma + v_exp_f32: exp2(x * log2(e) - m * log2(e)); callers keep m pre-multiplied ("m2")
constexpr float kLog2e = 1.4426950408889634f;
__device__ __forceinline__ float exp2_sub(float x, float m2) { return __builtin_amdgcn_exp2f(fmaf(x, kLog2e, -m2)); }

// Every in-loop operand is one 16-byte load per lane at (uniform tile base) + (constant lane offset): Q / K rows come
// from copies zero-padded to 16 features ([len][16]; the tensors themselves when d_qk = 16), Q^T / K^T from "columns"
// packings zero-padded to 32 feature rows, so the zero slots of the padded MFMA operands are read, not selected.
// 16 accumulator values of one lane (rows 8q + 4 kgrp + j) -> the two B fragments of the next MFMA's two K steps:
// step 0 = reduction indices 16 kgrp + 0..7, step 1 = 16 kgrp + 8..15 (see the header)
template <bool F16>
__device__ __forceinline__ void acc_to_b(const float (&v)[16], bf16x8 (&b)[2]) {
  unsigned p[4][2];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    p[q][0] = pack16x2<F16>(v[q * 4 + 0], v[q * 4 + 1]);
    p[q][1] = pack16x2<F16>(v[q * 4 + 2], v[q * 4 + 3]);
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
  b[0] = __builtin_bit_cast(bf16x8, o0);
  b[1] = __builtin_bit_cast(bf16x8, o1);
}

// stores a lane's 16 values of one 32-row block (rows 8q + 4 kgrp + j of column l31) as 16 consecutive 16-bit elements
// at dst + 16 kgrp (dst = the column's row of a [..][32-block] matrix): the conv epilogue's swap + two 16-byte stores
template <bool F16>
__device__ __forceinline__ void store_block(bf16* dst, int kgrp, const float (&v)[16]) {
  bf16x8 b[2];
  acc_to_b<F16>(v, b);
  *reinterpret_cast<bf16x8*>(dst + 16 * kgrp) = b[0];
  *reinterpret_cast<bf16x8*>(dst + 16 * kgrp + 8) = b[1];
}


// fragment-order copies -----------------------------------------------------------------------------------------------
// "rows" packing of x [rows_total][d]: fragment (rb, t) = rows 32 rb .. + 32 x features 16 t .. + 16 as an A (or B)
// operand with the FEATURE axis as K: dst[((rb * (d / 16) + t) * 64 + lane) * 8 + i] = x[32 rb + lane % 32][16 t + 8 (lane / 32) + i]
__global__ __launch_bounds__(256) void pack_rows_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst, int d,
                                                        int64_t vecs) {
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (v >= vecs) return;
  const int lane = (int)(v & 63), per = d >> 4;
  const int64_t frag = v >> 6, rb = frag / per;
  const int t = (int)(frag - rb * per);
  *reinterpret_cast<bf16x8*>(dst + v * 8) = load_frag(src + (rb * 32 + (lane & 31)) * d + 16 * t + 8 * (lane >> 5));
}

// "columns" packing of x [n][len][d]: fragment (img, db, pb, st) = features 32 db .. + 32 x positions of block pb as an A
// operand with the POSITION axis as K, in the order acc_to_b emits the matching B operand:
// dst[((((img * ndb + db) * (len / 32) + pb) * 2 + st) * 64 + lane) * 8 + i] = x[img][32 pb + 16 (lane / 32) + 8 st + i][32 db + lane % 32]
// with ndb = ceil(d / 32) and zeros for the feature rows >= d
__global__ __launch_bounds__(256) void pack_cols_kernel(const unsigned short* __restrict__ src, unsigned short* __restrict__ dst,
                                                        int len, int d, int64_t vecs) {
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (v >= vecs) return;
  const int lane = (int)(v & 63), st = (int)((v >> 6) & 1);
  int64_t r = v >> 7;
  const int nb = len >> 5, ndb = (d + 31) >> 5;
  const int pb = (int)(r % nb);
  r /= nb;
  const int db = (int)(r % ndb);
  const int64_t img = r / ndb;
  const int col = 32 * db + (lane & 31);
  typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
  u16x8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = 0;
  if (col < d) {
    const unsigned short* s0 = src + (img * len + 32 * pb + 16 * (lane >> 5) + 8 * st) * d + col;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = s0[(int64_t)i * d];
  }
  *reinterpret_cast<u16x8*>(dst + v * 8) = o;
}

// x [rows][8] -> [rows][16] with zeros in features 8..15 (the K padding of the score MFMAs for d_qk = 8)
__global__ __launch_bounds__(256) void pad16_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst, int64_t rows) {
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;      // one 16-byte half row of dst
  if (v >= 2 * rows) return;
  *reinterpret_cast<bf16x8*>(dst + v * 8) = (v & 1) ? zero_frag() : load_frag(src + (v >> 1) * 8);
}

// ------------------------------------------------------------------------------------------------
// forward: O[i] = sum_j softmax_j(q_i . k_j) v_j,  lse[i] = log sum_j exp(q_i . k_j)
// grid = (len / 128, n); 4 waves x 32 queries.  vpc = "columns" packing of V.
// ------------------------------------------------------------------------------------------------
template <int DVB>
struct FwdTile {
  bf16x8 kf, vf[DVB][2];
};

// The running maximum is allowed to go STALE by up to kStaleMax (natural-log units): the accumulators are rescaled only
// when a block's maximum exceeds it by more than that (a wave-uniform branch, taken a handful of times per query), so
// the probabilities are bounded by e^kStaleMax instead of 1 -- harmless in fp32 sums and in 16-bit P.  The row sums come
// from the MFMA unit too (an all-ones A operand against the same packed P), i.e. they are sums of the ROUNDED
// probabilities the numerator uses.
constexpr float kStaleMax = 6.f;

template <int DVB, bool F16>
__global__ __launch_bounds__(256) void flash_fwd_kernel(const bf16* __restrict__ q16, const bf16* __restrict__ k16,
                                                        const bf16* __restrict__ vpc, bf16* __restrict__ o,
                                                        float* __restrict__ lse, const FlashGeom g) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, kgrp = lane >> 5, l31 = lane & 31;
  const int img = blockIdx.y, nb = g.len >> 5;
  const int q0 = (blockIdx.x * 4 + wid) * 32;
  const unsigned lfeat = l31 * 16 + 8 * kgrp, lfrag = lane * 8;      // lane offsets: a [32][16] row block, a 1-KB fragment
  const bf16* ki = k16 + (size_t)img * g.len * 16;                   // uniform bases
  const bf16* vi = vpc + (size_t)img * g.dv * g.len;
  const bf16x8 qf = load_frag(q16 + ((size_t)img * g.len + q0) * 16 + lfeat);      // B operand of S^T, constant over the loop
  const bf16x8 ones = ones_frag<F16>();
  f32x16 acc[DVB], lacc, zero;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero[r] = lacc[r] = 0.f;
#pragma unroll
  for (int d = 0; d < DVB; ++d) acc[d] = zero;
  float m_run = -INFINITY, m2 = -INFINITY;      // the (stale) maximum and the same times log2(e)
  // operand schedule: the S^T operand of the NEXT block and this block's V^T fragments (needed only after the
  // softmax arithmetic) are requested at the top of a block, pinned there by the scheduling barrier
  // (two blocks per loop trip, the one-ahead registers ping-ponging between kf_a and kf_b: no copies)
  bf16x8 kf_a = load_frag(ki + lfeat), kf_b;
  auto tile = [&](int kb, const bf16x8& kf_cur, bf16x8& kf_next) {
    FwdTile<DVB> t;
    kf_next = load_frag(ki + (size_t)(kb + 1 < nb ? kb + 1 : kb) * 512 + lfeat);
    t.kf = kf_cur;
#pragma unroll
    for (int d = 0; d < DVB; ++d)
#pragma unroll
      for (int st = 0; st < 2; ++st) t.vf[d][st] = load_frag(vi + (((size_t)d * nb + kb) * 2 + st) * 512 + lfrag);
    __builtin_amdgcn_sched_barrier(0);
    const f32x16 s = mfma_32x32x16<F16>(t.kf, qf, zero);      // S^T[key][query]
    float mx = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const bool raise = mx > m_run + kStaleMax;      // always on the first block (m_run = -inf)
    if (__any(raise)) {
      const float m_new = raise ? mx : m_run;
      const float corr = m_run == -INFINITY ? 0.f : __builtin_amdgcn_exp2f((m_run - m_new) * kLog2e);
#pragma unroll
      for (int d = 0; d < DVB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[d][r] *= corr;
#pragma unroll
      for (int r = 0; r < 16; ++r) lacc[r] *= corr;
      m_run = m_new;
      m2 = m_new * kLog2e;
    }
    float p[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) p[r] = exp2_sub(s[r], m2);
    bf16x8 pb[2];
    acc_to_b<F16>(p, pb);
    lacc = mfma_32x32x16<F16>(ones, pb[0], lacc);      // every row: sum_j P[j][query]
    lacc = mfma_32x32x16<F16>(ones, pb[1], lacc);
#pragma unroll
    for (int d = 0; d < DVB; ++d) {
      acc[d] = mfma_32x32x16<F16>(t.vf[d][0], pb[0], acc[d]);
      acc[d] = mfma_32x32x16<F16>(t.vf[d][1], pb[1], acc[d]);
    }
  };
  for (int kb = 0; kb < nb; kb += 2) {      // len % 128 == 0: an even number of blocks
    tile(kb, kf_a, kf_b);
    tile(kb + 1, kf_b, kf_a);
  }
  const float l_run = lacc[0];
  const float inv = 1.f / l_run;
  bf16* orow = o + ((size_t)img * g.len + q0 + l31) * g.dv;
#pragma unroll
  for (int d = 0; d < DVB; ++d) {
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = acc[d][r] * inv;
    store_block<F16>(orow + 32 * d, kgrp, v);
  }
  if (kgrp == 0) lse[(size_t)img * g.len + q0 + l31] = m_run + __logf(l_run);
}

// dvec[i] = sum_d dO[i][d] * O[i][d]  (= rowsum(dP o P), the softmax backward's correction term) and lse2[i] = lse[i] *
// log2(e) (the backward's exp2 form); one thread per row
template <typename E>
__global__ void flash_rowdot_kernel(const E* __restrict__ d_o, const E* __restrict__ o, const float* __restrict__ lse,
                                    float* __restrict__ dvec, float* __restrict__ lse2, int64_t rows, int dv) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows) return;
  float a = 0.f;
  for (int c = 0; c < dv; c += 8) {
    const Vec16<E> x = ldv(d_o + i * dv + c), y = ldv(o + i * dv + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) a = fmaf(x.get(j), y.get(j), a);
  }
  dvec[i] = a;
  lse2[i] = lse[i] * kLog2e;
}

// stores rows [0, dk) of a transposed accumulator (row = 8 (r / 4) + 4 kgrp + r % 4, column = the lane's position) as
// dst[0 .. dk) of that position: 4 consecutive 16-bit values per stored quad
template <bool F16>
__device__ __forceinline__ void store_feat(bf16* dst, int dk, int kgrp, const f32x16& acc) {
  typedef __attribute__((ext_vector_type(2))) unsigned u2;
#pragma unroll
  for (int qd = 0; qd < 2; ++qd) {
    if (8 * qd < dk) {
      u2 pk;
      pk[0] = pack16x2<F16>(acc[4 * qd + 0], acc[4 * qd + 1]);
      pk[1] = pack16x2<F16>(acc[4 * qd + 2], acc[4 * qd + 3]);
      *reinterpret_cast<u2*>(dst + 8 * qd + 4 * kgrp) = pk;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward, queries on the lanes: dQ[i] = sum_j dS[i][j] k_j with dS = P o (dO V^T - D), P = exp(S - lse)
// vpr = "rows" packing of V, kt = K^T [n][dk][len]
// ------------------------------------------------------------------------------------------------
template <int KT>
struct BwdQTile {
  bf16x8 kf, vf[KT], ktf[2];
};

template <int DVB, bool F16>
__global__ __launch_bounds__(256) void flash_bwd_q_kernel(const bf16* __restrict__ q16, const bf16* __restrict__ k16,
                                                          const bf16* __restrict__ vpr, const bf16* __restrict__ kpc,
                                                          const bf16* __restrict__ d_o, const float* __restrict__ lse2,
                                                          const float* __restrict__ dvec, bf16* __restrict__ dq,
                                                          const FlashGeom g) {
  constexpr int KT = DVB * 2;      // K steps over d_v
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, kgrp = lane >> 5, l31 = lane & 31;
  const int img = blockIdx.y, nb = g.len >> 5;
  const int q0 = (blockIdx.x * 4 + wid) * 32;
  const size_t row = (size_t)img * g.len + q0 + l31;
  const unsigned lfeat = l31 * 16 + 8 * kgrp, lfrag = lane * 8;
  const bf16* ki = k16 + (size_t)img * g.len * 16;
  const bf16* vi = vpr + (size_t)img * g.len * g.dv;
  const bf16* kci = kpc + (size_t)img * g.len * 32;      // K^T fragments: [len / 32][2] of 1 KB
  const bf16x8 qf = load_frag(q16 + ((size_t)img * g.len + q0) * 16 + lfeat);
  const float lse_q = lse2[row], d_q = dvec[row];
  bf16x8 dof[KT];      // B operand of dP^T = V dO^T: dO[query][16 t + 8 kgrp + i]
#pragma unroll
  for (int t = 0; t < KT; ++t) dof[t] = load_frag(d_o + row * g.dv + 16 * t + 8 * kgrp);
  f32x16 zero, acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero[r] = acc[r] = 0.f;
  // operand schedule as in the forward: S^T / dP^T operands one block ahead, the K^T fragments of this block at its top
  struct Early {
    bf16x8 kf, vf[KT];
  } ea, eb;
  auto load_early = [&](Early& e, int kb) {
    e.kf = load_frag(ki + (size_t)kb * 512 + lfeat);
#pragma unroll
    for (int c = 0; c < KT; ++c) e.vf[c] = load_frag(vi + ((size_t)kb * KT + c) * 512 + lfrag);
  };
  load_early(ea, 0);
  auto tile = [&](int kb, const Early& cur, Early& next) {
    BwdQTile<KT> t;
    load_early(next, kb + 1 < nb ? kb + 1 : kb);
    t.kf = cur.kf;
#pragma unroll
    for (int c = 0; c < KT; ++c) t.vf[c] = cur.vf[c];
#pragma unroll
    for (int st = 0; st < 2; ++st) t.ktf[st] = load_frag(kci + ((size_t)kb * 2 + st) * 512 + lfrag);
    __builtin_amdgcn_sched_barrier(0);
    const f32x16 s = mfma_32x32x16<F16>(t.kf, qf, zero);      // S^T[key][query]
    f32x16 dp = zero;                                         // dP^T[key][query]
#pragma unroll
    for (int c = 0; c < KT; ++c) dp = mfma_32x32x16<F16>(t.vf[c], dof[c], dp);
    float ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) ds[r] = exp2_sub(s[r], lse_q) * (dp[r] - d_q);
    bf16x8 dsb[2];
    acc_to_b<F16>(ds, dsb);
    acc = mfma_32x32x16<F16>(t.ktf[0], dsb[0], acc);          // dQ^T[d_qk][query]
    acc = mfma_32x32x16<F16>(t.ktf[1], dsb[1], acc);
  };
  for (int kb = 0; kb < nb; kb += 2) {
    tile(kb, ea, eb);
    tile(kb + 1, eb, ea);
  }
  store_feat<F16>(dq + row * g.dk, g.dk, kgrp, acc);
}

// ------------------------------------------------------------------------------------------------
// backward, keys on the lanes: dV[j] = sum_i P[i][j] dO_i,  dK[j] = sum_i dS[i][j] q_i
// qt = Q^T [n][dk][len]; dopr / dopc = "rows" / "columns" packings of dO
// ------------------------------------------------------------------------------------------------
template <int DVB>
struct BwdKvTile {
  bf16x8 qfa, dofa[2 * DVB], dotf[DVB][2], qtf[2];
};

template <int DVB, bool F16>
__global__ __launch_bounds__(256) void flash_bwd_kv_kernel(const bf16* __restrict__ q16, const bf16* __restrict__ k16,
                                                           const bf16* __restrict__ v, const bf16* __restrict__ qpc,
                                                           const bf16* __restrict__ dopr, const bf16* __restrict__ dopc,
                                                           const float* __restrict__ lse2, const float* __restrict__ dvec,
                                                           bf16* __restrict__ dk_out, bf16* __restrict__ dv_out,
                                                           const FlashGeom g) {
  constexpr int KT = DVB * 2;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, kgrp = lane >> 5, l31 = lane & 31;
  const int img = blockIdx.y, nb = g.len >> 5;
  const int key0 = (blockIdx.x * 4 + wid) * 32;
  const size_t krow = (size_t)img * g.len + key0 + l31;
  const unsigned lfeat = l31 * 16 + 8 * kgrp, lfrag = lane * 8;
  const bf16* qi = q16 + (size_t)img * g.len * 16;
  const bf16* dri = dopr + (size_t)img * g.len * g.dv;
  const bf16* dci = dopc + (size_t)img * g.len * g.dv;
  const bf16* qci = qpc + (size_t)img * g.len * 32;      // Q^T fragments
  const float* lsei = lse2 + (size_t)img * g.len + 4 * kgrp;
  const float* dvi = dvec + (size_t)img * g.len + 4 * kgrp;
  const bf16x8 kfb = load_frag(k16 + ((size_t)img * g.len + key0) * 16 + lfeat);      // B of S = Q K^T
  bf16x8 vfb[KT];                                                                           // B of dP = dO V^T
#pragma unroll
  for (int t = 0; t < KT; ++t) vfb[t] = load_frag(v + krow * g.dv + 16 * t + 8 * kgrp);
  f32x16 zero, acck, accv[DVB];
#pragma unroll
  for (int r = 0; r < 16; ++r) zero[r] = acck[r] = 0.f;
#pragma unroll
  for (int d = 0; d < DVB; ++d) accv[d] = zero;
  // operand schedule: S / dP operands (Q rows, dO "rows" fragments) one block ahead; the dO^T / Q^T fragments, needed
  // after the exponentials, at the top of their own block
  struct Early {
    bf16x8 qfa, dofa[KT];
  } ea, eb;
  auto load_early = [&](Early& e, int qb) {
    e.qfa = load_frag(qi + (size_t)qb * 512 + lfeat);      // A of S: rows = queries
#pragma unroll
    for (int c = 0; c < KT; ++c) e.dofa[c] = load_frag(dri + ((size_t)qb * KT + c) * 512 + lfrag);
  };
  load_early(ea, 0);
  auto tile = [&](int qb, const Early& cur, Early& next) {
    BwdKvTile<DVB> t;
    load_early(next, qb + 1 < nb ? qb + 1 : qb);
    t.qfa = cur.qfa;
#pragma unroll
    for (int c = 0; c < KT; ++c) t.dofa[c] = cur.dofa[c];
#pragma unroll
    for (int d = 0; d < DVB; ++d)
#pragma unroll
      for (int st = 0; st < 2; ++st) t.dotf[d][st] = load_frag(dci + (((size_t)d * nb + qb) * 2 + st) * 512 + lfrag);
#pragma unroll
    for (int st = 0; st < 2; ++st) t.qtf[st] = load_frag(qci + ((size_t)qb * 2 + st) * 512 + lfrag);
    __builtin_amdgcn_sched_barrier(0);
    const f32x16 s = mfma_32x32x16<F16>(t.qfa, kfb, zero);      // S[query][key]
    f32x16 dp = zero;
#pragma unroll
    for (int c = 0; c < KT; ++c) dp = mfma_32x32x16<F16>(t.dofa[c], vfb[c], dp);      // dP[query][key]
    // per accumulator ROW = per query of the block: rows 8 j + 4 kgrp + 0..3, the same for all lanes of a half-wave
    float p[16], ds[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 lr = *reinterpret_cast<const float4*>(lsei + 32 * qb + 8 * j);
      const float4 dr = *reinterpret_cast<const float4*>(dvi + 32 * qb + 8 * j);
      const float lrv[4] = {lr.x, lr.y, lr.z, lr.w}, drv[4] = {dr.x, dr.y, dr.z, dr.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        p[4 * j + i] = exp2_sub(s[4 * j + i], lrv[i]);
        ds[4 * j + i] = p[4 * j + i] * (dp[4 * j + i] - drv[i]);
      }
    }
    bf16x8 pb[2], dsb[2];
    acc_to_b<F16>(p, pb);
    acc_to_b<F16>(ds, dsb);
#pragma unroll
    for (int d = 0; d < DVB; ++d) {
      accv[d] = mfma_32x32x16<F16>(t.dotf[d][0], pb[0], accv[d]);      // dV^T[d_v][key]
      accv[d] = mfma_32x32x16<F16>(t.dotf[d][1], pb[1], accv[d]);
    }
    acck = mfma_32x32x16<F16>(t.qtf[0], dsb[0], acck);                   // dK^T[d_qk][key]
    acck = mfma_32x32x16<F16>(t.qtf[1], dsb[1], acck);
  };
  for (int qb = 0; qb < nb; qb += 2) {
    tile(qb, ea, eb);
    tile(qb + 1, eb, ea);
  }
#pragma unroll
  for (int d = 0; d < DVB; ++d) {
    float vals[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) vals[r] = accv[d][r];
    store_block<F16>(dv_out + krow * g.dv + 32 * d, kgrp, vals);
  }
  store_feat<F16>(dk_out + krow * g.dk, g.dk, kgrp, acck);
}

// ------------------------------------------------------------------------------------------------
// SECOND ORDER: the backward of the first-order backward (the gradient penalty differentiates d pred / d x, i.e. the
// (dq, dk, dv) above as functions of (q, k, v, dO), image_generation.py:414-439).  Given the cotangents (aQ, aK, aV) of
// (dQ, dK, dV):
//   gP = dO V^T,  D = rowsum(P o gP),  gS = P o (gP - D)                       (the first-order quantities)
//   W = aQ K^T + Q aK^T,  E = rowsum(P o W),  T = P o (W - E)                  (T: adjoint of gP)
//   Y = dO aV^T,  X = Y + W o (gP - D) - E gP,  F = rowsum(P o X),  U = P o (X - F)   (X, U: adjoints of P, S)
//   adj Q = gS aK + U K,  adj K = gS^T aQ + U^T Q,  adj V = T^T dO,  adj dO = P aV + T V
// (checked against autograd in float64).  Three kernels in the first-order kernels' two orientations: a statistics
// pass (E, F = rowsum(P o Y) + rowsum(P o W o gP) - 2 D E per query), the query-side pass (adj Q, adj dO) and the
// key-side pass (adj K, adj V); every [N x N] quantity lives in one 32 x 32 MFMA accumulator at a time.
// ------------------------------------------------------------------------------------------------
template <int DVB, bool F16>
__global__ __launch_bounds__(256) void flash_bb_stats_kernel(const bf16* __restrict__ q16, const bf16* __restrict__ k16,
                                                             const bf16* __restrict__ aq16, const bf16* __restrict__ ak16,
                                                             const bf16* __restrict__ vpr, const bf16* __restrict__ avpr,
                                                             const bf16* __restrict__ d_o, const float* __restrict__ lse2,
                                                             const float* __restrict__ dvec, float* __restrict__ e_out,
                                                             float* __restrict__ f_out, const FlashGeom g) {
  constexpr int KT = DVB * 2;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, kgrp = lane >> 5, l31 = lane & 31;
  const int img = blockIdx.y, nb = g.len >> 5;
  const int q0 = (blockIdx.x * 4 + wid) * 32;
  const size_t row = (size_t)img * g.len + q0 + l31;
  const unsigned lfeat = l31 * 16 + 8 * kgrp, lfrag = lane * 8;
  const bf16* ki = k16 + (size_t)img * g.len * 16;
  const bf16* aki = ak16 + (size_t)img * g.len * 16;
  const bf16* vi = vpr + (size_t)img * g.len * g.dv;
  const bf16* avi = avpr + (size_t)img * g.len * g.dv;
  const bf16x8 qf = load_frag(q16 + ((size_t)img * g.len + q0) * 16 + lfeat);
  const bf16x8 aqf = load_frag(aq16 + ((size_t)img * g.len + q0) * 16 + lfeat);
  const float lse_q = lse2[row];
  bf16x8 gof[KT];      // B operand of gP^T = V dO^T and Y^T = aV dO^T
#pragma unroll
  for (int t = 0; t < KT; ++t) gof[t] = load_frag(d_o + row * g.dv + 16 * t + 8 * kgrp);
  f32x16 zero;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero[r] = 0.f;
  float e = 0.f, h = 0.f, jj = 0.f;
  for (int kb = 0; kb < nb; ++kb) {
    const bf16x8 kf = load_frag(ki + (size_t)kb * 512 + lfeat), akf = load_frag(aki + (size_t)kb * 512 + lfeat);
    bf16x8 vf[KT], avf[KT];
#pragma unroll
    for (int c = 0; c < KT; ++c) {
      vf[c] = load_frag(vi + ((size_t)kb * KT + c) * 512 + lfrag);
      avf[c] = load_frag(avi + ((size_t)kb * KT + c) * 512 + lfrag);
    }
    __builtin_amdgcn_sched_barrier(0);
    const f32x16 s = mfma_32x32x16<F16>(kf, qf, zero);                       // S^T[key][query]
    f32x16 w = mfma_32x32x16<F16>(kf, aqf, zero);                            // W^T = K aQ^T + aK Q^T
    w = mfma_32x32x16<F16>(akf, qf, w);
    f32x16 gp = zero, y = zero;
#pragma unroll
    for (int c = 0; c < KT; ++c) {
      gp = mfma_32x32x16<F16>(vf[c], gof[c], gp);                           // gP^T = V dO^T
      y = mfma_32x32x16<F16>(avf[c], gof[c], y);                            // Y^T = aV dO^T
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = exp2_sub(s[r], lse_q), pw = p * w[r];
      e += pw;
      h = fmaf(pw, gp[r], h);
      jj = fmaf(p, y[r], jj);
    }
  }
  e += __shfl_xor(e, 32, 64);
  h += __shfl_xor(h, 32, 64);
  jj += __shfl_xor(jj, 32, 64);
  if (kgrp == 0) {
    e_out[row] = e;
    f_out[row] = jj + h - 2.f * dvec[row] * e;
  }
}

// query side: adj Q = gS aK + U K,  adj dO = P aV + T V.  kpc / akpc / vpc / avpc: "columns" packings
template <int DVB, bool F16>
__global__ __launch_bounds__(256) void flash_bb_q_kernel(const bf16* __restrict__ q16, const bf16* __restrict__ k16,
                                                         const bf16* __restrict__ aq16, const bf16* __restrict__ ak16,
                                                         const bf16* __restrict__ vpr, const bf16* __restrict__ avpr,
                                                         const bf16* __restrict__ kpc, const bf16* __restrict__ akpc,
                                                         const bf16* __restrict__ vpc, const bf16* __restrict__ avpc,
                                                         const bf16* __restrict__ d_o, const float* __restrict__ lse2,
                                                         const float* __restrict__ dvec, const float* __restrict__ evec,
                                                         const float* __restrict__ fvec, bf16* __restrict__ adj_q,
                                                         bf16* __restrict__ adj_do, const FlashGeom g) {
  constexpr int KT = DVB * 2;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, kgrp = lane >> 5, l31 = lane & 31;
  const int img = blockIdx.y, nb = g.len >> 5;
  const int q0 = (blockIdx.x * 4 + wid) * 32;
  const size_t row = (size_t)img * g.len + q0 + l31;
  const unsigned lfeat = l31 * 16 + 8 * kgrp, lfrag = lane * 8;
  const bf16* ki = k16 + (size_t)img * g.len * 16;
  const bf16* aki = ak16 + (size_t)img * g.len * 16;
  const bf16* vi = vpr + (size_t)img * g.len * g.dv;
  const bf16* avi = avpr + (size_t)img * g.len * g.dv;
  const bf16* kci = kpc + (size_t)img * g.len * 32;
  const bf16* akci = akpc + (size_t)img * g.len * 32;
  const bf16* vci = vpc + (size_t)img * g.len * g.dv;
  const bf16* avci = avpc + (size_t)img * g.len * g.dv;
  const bf16x8 qf = load_frag(q16 + ((size_t)img * g.len + q0) * 16 + lfeat);
  const bf16x8 aqf = load_frag(aq16 + ((size_t)img * g.len + q0) * 16 + lfeat);
  const float lse_q = lse2[row], d_q = dvec[row], e_q = evec[row], f_q = fvec[row];
  bf16x8 gof[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t) gof[t] = load_frag(d_o + row * g.dv + 16 * t + 8 * kgrp);
  f32x16 zero, accq, accg[DVB];
#pragma unroll
  for (int r = 0; r < 16; ++r) zero[r] = 0.f;
  accq = zero;
#pragma unroll
  for (int d = 0; d < DVB; ++d) accg[d] = zero;
  for (int kb = 0; kb < nb; ++kb) {
    const bf16x8 kf = load_frag(ki + (size_t)kb * 512 + lfeat), akf = load_frag(aki + (size_t)kb * 512 + lfeat);
    bf16x8 vf[KT], avf[KT];
#pragma unroll
    for (int c = 0; c < KT; ++c) {
      vf[c] = load_frag(vi + ((size_t)kb * KT + c) * 512 + lfrag);
      avf[c] = load_frag(avi + ((size_t)kb * KT + c) * 512 + lfrag);
    }
    __builtin_amdgcn_sched_barrier(0);
    const f32x16 s = mfma_32x32x16<F16>(kf, qf, zero);
    f32x16 w = mfma_32x32x16<F16>(kf, aqf, zero);
    w = mfma_32x32x16<F16>(akf, qf, w);
    f32x16 gp = zero, y = zero;
#pragma unroll
    for (int c = 0; c < KT; ++c) {
      gp = mfma_32x32x16<F16>(vf[c], gof[c], gp);
      y = mfma_32x32x16<F16>(avf[c], gof[c], y);
    }
    // the post-softmax operands of this block: requested here, needed after the element-wise part
    bf16x8 kcf[2], akcf[2], vcf[DVB][2], avcf[DVB][2];
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      kcf[st] = load_frag(kci + ((size_t)kb * 2 + st) * 512 + lfrag);
      akcf[st] = load_frag(akci + ((size_t)kb * 2 + st) * 512 + lfrag);
#pragma unroll
      for (int d = 0; d < DVB; ++d) {
        vcf[d][st] = load_frag(vci + (((size_t)d * nb + kb) * 2 + st) * 512 + lfrag);
        avcf[d][st] = load_frag(avci + (((size_t)d * nb + kb) * 2 + st) * 512 + lfrag);
      }
    }
    float pv[16], gs[16], tv[16], uv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = exp2_sub(s[r], lse_q), gpd = gp[r] - d_q;
      pv[r] = p;
      gs[r] = p * gpd;
      tv[r] = p * (w[r] - e_q);
      const float x = fmaf(w[r], gpd, y[r]) - e_q * gp[r];
      uv[r] = p * (x - f_q);
    }
    bf16x8 pb[2], gsb[2], tb[2], ub[2];
    acc_to_b<F16>(pv, pb);
    acc_to_b<F16>(gs, gsb);
    acc_to_b<F16>(tv, tb);
    acc_to_b<F16>(uv, ub);
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      accq = mfma_32x32x16<F16>(akcf[st], gsb[st], accq);      // adj Q^T += aK^T gS^T
      accq = mfma_32x32x16<F16>(kcf[st], ub[st], accq);        //          + K^T U^T
#pragma unroll
      for (int d = 0; d < DVB; ++d) {
        accg[d] = mfma_32x32x16<F16>(avcf[d][st], pb[st], accg[d]);      // adj dO^T += aV^T P^T
        accg[d] = mfma_32x32x16<F16>(vcf[d][st], tb[st], accg[d]);       //           + V^T T^T
      }
    }
  }
  store_feat<F16>(adj_q + row * g.dk, g.dk, kgrp, accq);
#pragma unroll
  for (int d = 0; d < DVB; ++d) {
    float vals[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) vals[r] = accg[d][r];
    store_block<F16>(adj_do + row * g.dv + 32 * d, kgrp, vals);
  }
}

// key side: adj K = gS^T aQ + U^T Q,  adj V = T^T dO.  qpc / aqpc / dopc: "columns" packings, dopr: "rows" packing of dO
template <int DVB, bool F16>
__global__ __launch_bounds__(256) void flash_bb_kv_kernel(const bf16* __restrict__ q16, const bf16* __restrict__ k16,
                                                          const bf16* __restrict__ aq16, const bf16* __restrict__ ak16,
                                                          const bf16* __restrict__ v, const bf16* __restrict__ av,
                                                          const bf16* __restrict__ qpc, const bf16* __restrict__ aqpc,
                                                          const bf16* __restrict__ dopr, const bf16* __restrict__ dopc,
                                                          const float* __restrict__ lse2, const float* __restrict__ dvec,
                                                          const float* __restrict__ evec, const float* __restrict__ fvec,
                                                          bf16* __restrict__ adj_k, bf16* __restrict__ adj_v,
                                                          const FlashGeom g) {
  constexpr int KT = DVB * 2;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, kgrp = lane >> 5, l31 = lane & 31;
  const int img = blockIdx.y, nb = g.len >> 5;
  const int key0 = (blockIdx.x * 4 + wid) * 32;
  const size_t krow = (size_t)img * g.len + key0 + l31;
  const unsigned lfeat = l31 * 16 + 8 * kgrp, lfrag = lane * 8;
  const bf16* qi = q16 + (size_t)img * g.len * 16;
  const bf16* aqi = aq16 + (size_t)img * g.len * 16;
  const bf16* dri = dopr + (size_t)img * g.len * g.dv;
  const bf16* dci = dopc + (size_t)img * g.len * g.dv;
  const bf16* qci = qpc + (size_t)img * g.len * 32;
  const bf16* aqci = aqpc + (size_t)img * g.len * 32;
  const size_t soff = (size_t)img * g.len + 4 * kgrp;
  const bf16x8 kfb = load_frag(k16 + ((size_t)img * g.len + key0) * 16 + lfeat);      // B operands: this wave's keys
  const bf16x8 akfb = load_frag(ak16 + ((size_t)img * g.len + key0) * 16 + lfeat);
  bf16x8 vfb[KT], avfb[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    vfb[t] = load_frag(v + krow * g.dv + 16 * t + 8 * kgrp);
    avfb[t] = load_frag(av + krow * g.dv + 16 * t + 8 * kgrp);
  }
  f32x16 zero, acck, accv[DVB];
#pragma unroll
  for (int r = 0; r < 16; ++r) zero[r] = 0.f;
  acck = zero;
#pragma unroll
  for (int d = 0; d < DVB; ++d) accv[d] = zero;
  for (int qb = 0; qb < nb; ++qb) {
    const bf16x8 qfa = load_frag(qi + (size_t)qb * 512 + lfeat), aqfa = load_frag(aqi + (size_t)qb * 512 + lfeat);
    bf16x8 gofa[KT];
#pragma unroll
    for (int c = 0; c < KT; ++c) gofa[c] = load_frag(dri + ((size_t)qb * KT + c) * 512 + lfrag);
    __builtin_amdgcn_sched_barrier(0);
    const f32x16 s = mfma_32x32x16<F16>(qfa, kfb, zero);             // S[query][key]
    f32x16 w = mfma_32x32x16<F16>(aqfa, kfb, zero);                  // W = aQ K^T + Q aK^T
    w = mfma_32x32x16<F16>(qfa, akfb, w);
    f32x16 gp = zero, y = zero;
#pragma unroll
    for (int c = 0; c < KT; ++c) {
      gp = mfma_32x32x16<F16>(gofa[c], vfb[c], gp);                  // gP = dO V^T
      y = mfma_32x32x16<F16>(gofa[c], avfb[c], y);                   // Y = dO aV^T
    }
    bf16x8 qcf[2], aqcf[2], gocf[DVB][2];
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      qcf[st] = load_frag(qci + ((size_t)qb * 2 + st) * 512 + lfrag);
      aqcf[st] = load_frag(aqci + ((size_t)qb * 2 + st) * 512 + lfrag);
#pragma unroll
      for (int d = 0; d < DVB; ++d) gocf[d][st] = load_frag(dci + (((size_t)d * nb + qb) * 2 + st) * 512 + lfrag);
    }
    // per accumulator ROW = per query of the block: rows 8 j + 4 kgrp + 0..3
    float gs[16], tv[16], uv[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 lr = *reinterpret_cast<const float4*>(lse2 + soff + 32 * qb + 8 * j);
      const float4 dr = *reinterpret_cast<const float4*>(dvec + soff + 32 * qb + 8 * j);
      const float4 er = *reinterpret_cast<const float4*>(evec + soff + 32 * qb + 8 * j);
      const float4 fr = *reinterpret_cast<const float4*>(fvec + soff + 32 * qb + 8 * j);
      const float lrv[4] = {lr.x, lr.y, lr.z, lr.w}, drv[4] = {dr.x, dr.y, dr.z, dr.w};
      const float erv[4] = {er.x, er.y, er.z, er.w}, frv[4] = {fr.x, fr.y, fr.z, fr.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 4 * j + i;
        const float p = exp2_sub(s[r], lrv[i]), gpd = gp[r] - drv[i];
        gs[r] = p * gpd;
        tv[r] = p * (w[r] - erv[i]);
        const float x = fmaf(w[r], gpd, y[r]) - erv[i] * gp[r];
        uv[r] = p * (x - frv[i]);
      }
    }
    bf16x8 gsb[2], tb[2], ub[2];
    acc_to_b<F16>(gs, gsb);
    acc_to_b<F16>(tv, tb);
    acc_to_b<F16>(uv, ub);
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      acck = mfma_32x32x16<F16>(aqcf[st], gsb[st], acck);      // adj K^T += aQ^T gS
      acck = mfma_32x32x16<F16>(qcf[st], ub[st], acck);        //          + Q^T U
#pragma unroll
      for (int d = 0; d < DVB; ++d) accv[d] = mfma_32x32x16<F16>(gocf[d][st], tb[st], accv[d]);      // adj V^T += dO^T T
    }
  }
  store_feat<F16>(adj_k + krow * g.dk, g.dk, kgrp, acck);
#pragma unroll
  for (int d = 0; d < DVB; ++d) {
    float vals[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) vals[r] = accv[d][r];
    store_block<F16>(adj_v + krow * g.dv + 32 * d, kgrp, vals);
  }
}

// [n][rows][cols] -> [n][cols][rows], 16-bit elements, 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void transpose16_kernel(const unsigned short* __restrict__ src, unsigned short* __restrict__ dst,
                                                          int rows, int cols) {
  __shared__ unsigned short tile[32][33];
  const size_t base = (size_t)blockIdx.z * rows * cols;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
  for (int i = ty; i < 32; i += 8)
    if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = src[base + (size_t)(r0 + i) * cols + c0 + tx];
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (c0 + i < cols && r0 + tx < rows) dst[base + (size_t)(c0 + i) * rows + r0 + tx] = tile[tx][i];
}

}  // namespace

namespace {

inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

struct FlashWorkspace {      // offsets into the caller's workspace
  size_t q16, k16, v_pc, v_pr, do_pr, do_pc, q_pc, k_pc, dvec, lse2;
  size_t aq16, ak16, av_pr, av_pc, aq_pc, ak_pc, evec, fvec;      // second order only
  size_t total;
};

// pass: 0 = forward, 1 = first-order backward, 2 = second-order backward
FlashWorkspace flash_workspace(int n, int len, int dk, int dv, int pass) {
  const size_t big = align256((size_t)n * len * dv * 2), pad = align256((size_t)n * len * 16 * 2);
  const size_t cols = align256((size_t)n * len * 32 * 2), vec = align256((size_t)n * len * 4);
  FlashWorkspace w = {};
  size_t at = 0;
  auto take = [&](size_t bytes) {
    const size_t o = at;
    at += bytes;
    return o;
  };
  w.q16 = take(dk == 8 ? pad : 0);      // d_qk = 16: the tensors themselves
  w.k16 = take(dk == 8 ? pad : 0);
  if (pass == 0) {
    w.v_pc = take(big);
  } else {
    w.v_pr = take(big);
    w.do_pr = take(big);
    w.do_pc = take(big);
    w.q_pc = take(cols);
    w.k_pc = take(cols);
    w.dvec = take(vec);
    w.lse2 = take(vec);
  }
  if (pass == 2) {
    w.aq16 = take(dk == 8 ? pad : 0);
    w.ak16 = take(dk == 8 ? pad : 0);
    w.v_pc = take(big);
    w.av_pr = take(big);
    w.av_pc = take(big);
    w.aq_pc = take(cols);
    w.ak_pc = take(cols);
    w.evec = take(vec);
    w.fvec = take(vec);
  }
  w.total = at;
  return w;
}

void launch_pack_rows(const void* src, void* dst, int64_t rows, int d, hipStream_t s) {
  const int64_t vecs = rows * d / 8;
  hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)((vecs + 255) / 256)), dim3(256), 0, s, (const bf16*)src, (bf16*)dst, d, vecs);
}

void launch_pack_cols(const void* src, void* dst, int n, int len, int d, hipStream_t s) {
  const int64_t vecs = (int64_t)n * len * ((d + 31) / 32 * 32) / 8;
  hipLaunchKernelGGL(pack_cols_kernel, dim3((unsigned)((vecs + 255) / 256)), dim3(256), 0, s, (const unsigned short*)src,
                     (unsigned short*)dst, len, d, vecs);
}

// the [rows][16] form of a [rows][dk] tensor: the tensor itself for d_qk = 16, else a zero-padded copy in the workspace
const bf16* feat16(const void* x, void* pad, int64_t rows, int dk, hipStream_t s) {
  if (dk == 16) return (const bf16*)x;
  hipLaunchKernelGGL(pad16_kernel, dim3((unsigned)((2 * rows + 255) / 256)), dim3(256), 0, s, (const bf16*)x, (bf16*)pad, rows);
  return (const bf16*)pad;
}

void launch_transpose(const void* src, void* dst, int batch, int rows, int cols, hipStream_t s) {
  hipLaunchKernelGGL(transpose16_kernel, dim3((cols + 31) / 32, (rows + 31) / 32, batch), dim3(256), 0, s,
                     (const unsigned short*)src, (unsigned short*)dst, rows, cols);
}

}  // namespace

extern "C" {

int tg_transpose16(const void* src, void* dst, int batch, int rows, int cols, void* stream) {
  TG_CHECK(src && dst && batch > 0 && rows > 0 && cols > 0, TG_EINVAL, "tg_transpose16: bad arguments");
  launch_transpose(src, dst, batch, rows, cols, (hipStream_t)stream);
  TG_LAUNCH_CHECK("tg_transpose16");
  return TG_OK;
}

int tg_flash_attention_supported(int len, int dk, int dv) {
  return (len % 128 == 0 && (dk == 8 || dk == 16) && (dv == 64 || dv == 128 || dv == 256)) ? 1 : 0;
}

int64_t tg_flash_attention_workspace_bytes(int n, int len, int dk, int dv, int backward) {
  if (n <= 0 || !tg_flash_attention_supported(len, dk, dv)) return 0;
  return (int64_t)flash_workspace(n, len, dk, dv, backward < 0 ? 0 : (backward > 2 ? 2 : backward)).total;
}

int tg_flash_attention_fwd(const void* q, const void* k, const void* v, void* o, float* lse, void* workspace, int n, int len,
                           int dk, int dv, int dtype, void* stream) {
  TG_CHECK(q && k && v && o && lse && workspace && n > 0, TG_EINVAL, "tg_flash_attention_fwd: bad arguments");
  TG_CHECK(tg_flash_attention_supported(len, dk, dv), TG_ENOSUP,
           "tg_flash_attention_fwd: len %% 128 == 0, d_qk in {8, 16}, d_v in {64, 128, 256} (got %d, %d, %d)", len, dk, dv);
  TG_CHECK(dtype == TG_BF16 || dtype == TG_F16, TG_ENOSUP, "tg_flash_attention_fwd: 16-bit storage only");
  FlashGeom g;
  g.n = n; g.len = len; g.dk = dk; g.dv = dv;
  const dim3 grid(len / 128, n);
  hipStream_t s = (hipStream_t)stream;
  const FlashWorkspace w = flash_workspace(n, len, dk, dv, 0);
  char* ws = (char*)workspace;
  const int64_t rows = (int64_t)n * len;
  const bf16 *q16 = feat16(q, ws + w.q16, rows, dk, s), *k16 = feat16(k, ws + w.k16, rows, dk, s);
  launch_pack_cols(v, ws + w.v_pc, n, len, dv, s);
  const bool f16 = dtype == TG_F16;
#define TG_FL(DVB_, F16_)                                                                                               \
  hipLaunchKernelGGL((flash_fwd_kernel<DVB_, F16_>), grid, dim3(256), 0, s, q16, k16, (const bf16*)(ws + w.v_pc), (bf16*)o, \
                     lse, g)
#define TG_FL2(DVB_)                                                                                                    \
  do {                                                                                                                  \
    if (f16) TG_FL(DVB_, true);                                                                                         \
    else TG_FL(DVB_, false);                                                                                            \
  } while (0)
  if (dv == 64) TG_FL2(2);
  else if (dv == 128) TG_FL2(4);
  else TG_FL2(8);
#undef TG_FL2
#undef TG_FL
  TG_LAUNCH_CHECK("tg_flash_attention_fwd");
  return TG_OK;
}

int tg_flash_attention_bwd(const void* q, const void* k, const void* v, const void* d_o, const void* o, const float* lse,
                           void* workspace, void* dq, void* dk_out, void* dv_out, int n, int len, int dk, int dv, int dtype,
                           void* stream) {
  TG_CHECK(q && k && v && d_o && o && lse && workspace && dq && dk_out && dv_out && n > 0, TG_EINVAL,
           "tg_flash_attention_bwd: bad arguments");
  TG_CHECK(tg_flash_attention_supported(len, dk, dv) && dv <= 128, TG_ENOSUP,
           "tg_flash_attention_bwd: len %% 128 == 0, d_qk in {8, 16}, d_v in {64, 128} (got %d, %d, %d)", len, dk, dv);
  TG_CHECK(dtype == TG_BF16 || dtype == TG_F16, TG_ENOSUP, "tg_flash_attention_bwd: 16-bit storage only");
  FlashGeom g;
  g.n = n; g.len = len; g.dk = dk; g.dv = dv;
  hipStream_t s = (hipStream_t)stream;
  const FlashWorkspace w = flash_workspace(n, len, dk, dv, 1);
  char* ws = (char*)workspace;
  const bf16 *v_pr = (const bf16*)(ws + w.v_pr), *do_pr = (const bf16*)(ws + w.do_pr), *do_pc = (const bf16*)(ws + w.do_pc);
  const bf16 *q_pc = (const bf16*)(ws + w.q_pc), *k_pc = (const bf16*)(ws + w.k_pc);
  float *dvec = (float*)(ws + w.dvec), *lse2 = (float*)(ws + w.lse2);
  const int64_t rows = (int64_t)n * len;
  const bf16 *q16 = feat16(q, ws + w.q16, rows, dk, s), *k16 = feat16(k, ws + w.k16, rows, dk, s);
  launch_pack_rows(v, ws + w.v_pr, rows, dv, s);
  launch_pack_rows(d_o, ws + w.do_pr, rows, dv, s);
  launch_pack_cols(d_o, ws + w.do_pc, n, len, dv, s);
  launch_pack_cols(q, ws + w.q_pc, n, len, dk, s);
  launch_pack_cols(k, ws + w.k_pc, n, len, dk, s);
  if (dtype == TG_F16)
    hipLaunchKernelGGL(flash_rowdot_kernel<f16>, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, (const f16*)d_o,
                       (const f16*)o, lse, dvec, lse2, rows, dv);
  else
    hipLaunchKernelGGL(flash_rowdot_kernel<bf16>, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, (const bf16*)d_o,
                       (const bf16*)o, lse, dvec, lse2, rows, dv);
  const dim3 grid(len / 128, n);
  const bool f16 = dtype == TG_F16;
#define TG_FLB(DVB_, F16_)                                                                                               \
  do {                                                                                                                   \
    hipLaunchKernelGGL((flash_bwd_q_kernel<DVB_, F16_>), grid, dim3(256), 0, s, q16, k16, v_pr, k_pc, (const bf16*)d_o,   \
                       lse2, dvec, (bf16*)dq, g);                                                                         \
    hipLaunchKernelGGL((flash_bwd_kv_kernel<DVB_, F16_>), grid, dim3(256), 0, s, q16, k16, (const bf16*)v, q_pc, do_pr,   \
                       do_pc, lse2, dvec, (bf16*)dk_out, (bf16*)dv_out, g);                                               \
  } while (0)
#define TG_FLB2(DVB_)                                                                                                    \
  do {                                                                                                                   \
    if (f16) TG_FLB(DVB_, true);                                                                                         \
    else TG_FLB(DVB_, false);                                                                                            \
  } while (0)
  if (dv == 64) TG_FLB2(2);
  else TG_FLB2(4);
#undef TG_FLB2
#undef TG_FLB
  TG_LAUNCH_CHECK("tg_flash_attention_bwd");
  return TG_OK;
}

int tg_flash_attention_bwd_bwd(const void* q, const void* k, const void* v, const void* d_o, const void* o, const float* lse,
                               const void* a_q, const void* a_k, const void* a_v, void* workspace, void* adj_q, void* adj_k,
                               void* adj_v, void* adj_do, int n, int len, int dk, int dv, int dtype, void* stream) {
  TG_CHECK(q && k && v && d_o && o && lse && a_q && a_k && a_v && workspace && adj_q && adj_k && adj_v && adj_do && n > 0,
           TG_EINVAL, "tg_flash_attention_bwd_bwd: bad arguments");
  TG_CHECK(tg_flash_attention_supported(len, dk, dv) && dv <= 128, TG_ENOSUP,
           "tg_flash_attention_bwd_bwd: len %% 128 == 0, d_qk in {8, 16}, d_v in {64, 128} (got %d, %d, %d)", len, dk, dv);
  TG_CHECK(dtype == TG_BF16 || dtype == TG_F16, TG_ENOSUP, "tg_flash_attention_bwd_bwd: 16-bit storage only");
  FlashGeom g;
  g.n = n; g.len = len; g.dk = dk; g.dv = dv;
  hipStream_t s = (hipStream_t)stream;
  const FlashWorkspace w = flash_workspace(n, len, dk, dv, 2);
  char* ws = (char*)workspace;
  auto at = [&](size_t off) { return (const bf16*)(ws + off); };
  float *dvec = (float*)(ws + w.dvec), *lse2 = (float*)(ws + w.lse2), *evec = (float*)(ws + w.evec), *fvec = (float*)(ws + w.fvec);
  const int64_t rows = (int64_t)n * len;
  const bf16 *q16 = feat16(q, ws + w.q16, rows, dk, s), *k16 = feat16(k, ws + w.k16, rows, dk, s);
  const bf16 *aq16 = feat16(a_q, ws + w.aq16, rows, dk, s), *ak16 = feat16(a_k, ws + w.ak16, rows, dk, s);
  launch_pack_rows(v, ws + w.v_pr, rows, dv, s);
  launch_pack_rows(a_v, ws + w.av_pr, rows, dv, s);
  launch_pack_rows(d_o, ws + w.do_pr, rows, dv, s);
  launch_pack_cols(v, ws + w.v_pc, n, len, dv, s);
  launch_pack_cols(a_v, ws + w.av_pc, n, len, dv, s);
  launch_pack_cols(d_o, ws + w.do_pc, n, len, dv, s);
  launch_pack_cols(q, ws + w.q_pc, n, len, dk, s);
  launch_pack_cols(k, ws + w.k_pc, n, len, dk, s);
  launch_pack_cols(a_q, ws + w.aq_pc, n, len, dk, s);
  launch_pack_cols(a_k, ws + w.ak_pc, n, len, dk, s);
  if (dtype == TG_F16)
    hipLaunchKernelGGL(flash_rowdot_kernel<f16>, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, (const f16*)d_o,
                       (const f16*)o, lse, dvec, lse2, rows, dv);
  else
    hipLaunchKernelGGL(flash_rowdot_kernel<bf16>, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, (const bf16*)d_o,
                       (const bf16*)o, lse, dvec, lse2, rows, dv);
  const dim3 grid(len / 128, n);
#define TG_FBB(DVB_, F16_)                                                                                               \
  do {                                                                                                                   \
    hipLaunchKernelGGL((flash_bb_stats_kernel<DVB_, F16_>), grid, dim3(256), 0, s, q16, k16, aq16, ak16, at(w.v_pr),      \
                       at(w.av_pr), (const bf16*)d_o, lse2, dvec, evec, fvec, g);                                         \
    hipLaunchKernelGGL((flash_bb_q_kernel<DVB_, F16_>), grid, dim3(256), 0, s, q16, k16, aq16, ak16, at(w.v_pr),          \
                       at(w.av_pr), at(w.k_pc), at(w.ak_pc), at(w.v_pc), at(w.av_pc), (const bf16*)d_o, lse2, dvec, evec, \
                       fvec, (bf16*)adj_q, (bf16*)adj_do, g);                                                             \
    hipLaunchKernelGGL((flash_bb_kv_kernel<DVB_, F16_>), grid, dim3(256), 0, s, q16, k16, aq16, ak16, (const bf16*)v,     \
                       (const bf16*)a_v, at(w.q_pc), at(w.aq_pc), at(w.do_pr), at(w.do_pc), lse2, dvec, evec, fvec,       \
                       (bf16*)adj_k, (bf16*)adj_v, g);                                                                    \
  } while (0)
  const bool f16 = dtype == TG_F16;
  if (dv == 64 && f16) TG_FBB(2, true);
  else if (dv == 64) TG_FBB(2, false);
  else if (f16) TG_FBB(4, true);
  else TG_FBB(4, false);
#undef TG_FBB
  TG_LAUNCH_CHECK("tg_flash_attention_bwd_bwd");
  return TG_OK;
}

}  // extern "C"
