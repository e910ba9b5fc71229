// SAGAN self-attention (libs/self_attention.py:24-70: s = f g^T over the h*w positions, beta = softmax(s), o = beta h)
// without the [N x N] map: flash-style forward and first-order backward on the MFMA units.
//
// Shapes of the layer: N = h*w positions (4096 at 64x64), d_qk = c/8 (8 or 16), d_v = c (64..256).  All three kernels
// put the SOFTMAX ROWS' OWNERS ON THE LANES so that no reduction crosses lanes:
//   forward / dQ kernel: a wave owns 32 queries; S^T = K Q^T is one 32x32x16 MFMA per 32-key block (A = 32 keys x
//     d_qk padded to 16, B = d_qk x 32 queries), so lane l31 holds the scores of ITS query against 16 keys and its
//     partner lane l31 + 32 the other 16: the running max / sum are lane-local plus one half-wave exchange;
//   dK / dV kernel: a wave owns 32 keys, S = Q K^T tile per 32-query block, the per-query quantities (log-sum-exp, D)
//     are per accumulator ROW there and are read as broadcast loads.
// P (or dS) goes back into the next MFMA as the B operand after packing to 16 bit and the v_permlane32_swap of the conv
// epilogues: afterwards a lane of half kgrp holds 8 + 8 consecutive reduction indices (16 kgrp + 0..7 and + 8..15),
// which is the order the A operands (V^T, K^T, Q^T, dO^T rows from TRANSPOSED copies, 16-byte loads) are fetched in.
// Operands come straight from L2 (every workgroup of an image re-reads the same K / V: 0.5 MB per image), no LDS.
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

// rows [row0, row0 + 32) x features [8 kgrp, 8 kgrp + 8) of a [len][dk] matrix: an A (or B) fragment with the feature
// axis as K; feature slots >= dk are zero (d_qk = 8: the upper half-wave)
__device__ __forceinline__ bf16x8 feat_frag(const bf16* m, int row, int dk, int kgrp) {
  return 8 * kgrp < dk ? load_frag(m + (size_t)row * dk + 8 * kgrp) : zero_frag();
}

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

// ------------------------------------------------------------------------------------------------
// forward: O[i] = sum_j softmax_j(q_i . k_j) v_j,  lse[i] = log sum_j exp(q_i . k_j)
// grid = (len / 128, n); 4 waves x 32 queries.  vt = V^T [n][dv][len].
// ------------------------------------------------------------------------------------------------
template <int DVB, bool F16>
__global__ __launch_bounds__(256) void flash_fwd_kernel(const bf16* __restrict__ q, const bf16* __restrict__ k,
                                                        const bf16* __restrict__ vt, bf16* __restrict__ o,
                                                        float* __restrict__ lse, const FlashGeom g) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, kgrp = lane >> 5, l31 = lane & 31;
  const int img = blockIdx.y;
  const int q0 = (blockIdx.x * 4 + wid) * 32;
  const bf16* qi = q + (size_t)img * g.len * g.dk;
  const bf16* ki = k + (size_t)img * g.len * g.dk;
  const bf16* vti = vt + (size_t)img * g.dv * g.len;
  const bf16x8 qf = feat_frag(qi, q0 + l31, g.dk, kgrp);      // B operand of S^T, constant over the loop
  f32x16 acc[DVB];
#pragma unroll
  for (int d = 0; d < DVB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  f32x16 zero;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero[r] = 0.f;
  for (int k0 = 0; k0 < g.len; k0 += 32) {
    const bf16x8 kf = feat_frag(ki, k0 + l31, g.dk, kgrp);
    bf16x8 vf[DVB][2];
#pragma unroll
    for (int d = 0; d < DVB; ++d)
#pragma unroll
      for (int st = 0; st < 2; ++st) vf[d][st] = load_frag(vti + (size_t)(32 * d + l31) * g.len + k0 + 16 * kgrp + 8 * st);
    const f32x16 s = mfma_32x32x16<F16>(kf, qf, zero);      // S^T[key][query]
    float mx = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float corr = __expf(m_run - m_new);      // exp(-inf) = 0 on the first block
    float p[16], ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      p[r] = __expf(s[r] - m_new);
      ps += p[r];
    }
    ps += __shfl_xor(ps, 32, 64);
    l_run = l_run * corr + ps;
    m_run = m_new;
    bf16x8 pb[2];
    acc_to_b<F16>(p, pb);
#pragma unroll
    for (int d = 0; d < DVB; ++d) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[d][r] *= corr;
      acc[d] = mfma_32x32x16<F16>(vf[d][0], pb[0], acc[d]);
      acc[d] = mfma_32x32x16<F16>(vf[d][1], pb[1], acc[d]);
    }
  }
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

// dvec[i] = sum_d dO[i][d] * O[i][d]  (= rowsum(dP o P), the softmax backward's correction term); one thread per row
template <typename E>
__global__ void flash_rowdot_kernel(const E* __restrict__ d_o, const E* __restrict__ o, float* __restrict__ dvec,
                                    int64_t rows, int dv) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows) return;
  float a = 0.f;
  for (int c = 0; c < dv; c += 8) {
    const Vec16<E> x = ldv(d_o + i * dv + c), y = ldv(o + i * dv + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) a = fmaf(x.get(j), y.get(j), a);
  }
  dvec[i] = a;
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
// kt = K^T [n][dk][len]
// ------------------------------------------------------------------------------------------------
template <int DVB, bool F16>
__global__ __launch_bounds__(256) void flash_bwd_q_kernel(const bf16* __restrict__ q, const bf16* __restrict__ k,
                                                          const bf16* __restrict__ v, const bf16* __restrict__ kt,
                                                          const bf16* __restrict__ d_o, const float* __restrict__ lse,
                                                          const float* __restrict__ dvec, bf16* __restrict__ dq,
                                                          const FlashGeom g) {
  constexpr int KT = DVB * 2;      // K steps over d_v
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, kgrp = lane >> 5, l31 = lane & 31;
  const int img = blockIdx.y;
  const int q0 = (blockIdx.x * 4 + wid) * 32;
  const size_t row = (size_t)img * g.len + q0 + l31;
  const bf16* ki = k + (size_t)img * g.len * g.dk;
  const bf16* vi = v + (size_t)img * g.len * g.dv;
  const bf16* kti = kt + (size_t)img * g.dk * g.len;
  const bf16x8 qf = feat_frag(q + (size_t)img * g.len * g.dk, q0 + l31, g.dk, kgrp);
  const float lse_q = lse[row], d_q = dvec[row];
  bf16x8 dof[KT];      // B operand of dP^T = V dO^T: dO[query][16 t + 8 kgrp + i]
#pragma unroll
  for (int t = 0; t < KT; ++t) dof[t] = load_frag(d_o + row * g.dv + 16 * t + 8 * kgrp);
  f32x16 zero, acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero[r] = acc[r] = 0.f;
  for (int k0 = 0; k0 < g.len; k0 += 32) {
    const bf16x8 kf = feat_frag(ki, k0 + l31, g.dk, kgrp);
    bf16x8 vf[KT], ktf[2];
#pragma unroll
    for (int t = 0; t < KT; ++t) vf[t] = load_frag(vi + (size_t)(k0 + l31) * g.dv + 16 * t + 8 * kgrp);
#pragma unroll
    for (int st = 0; st < 2; ++st)
      ktf[st] = l31 < g.dk ? load_frag(kti + (size_t)l31 * g.len + k0 + 16 * kgrp + 8 * st) : zero_frag();
    const f32x16 s = mfma_32x32x16<F16>(kf, qf, zero);      // S^T[key][query]
    f32x16 dp = zero;                                       // dP^T[key][query]
#pragma unroll
    for (int t = 0; t < KT; ++t) dp = mfma_32x32x16<F16>(vf[t], dof[t], dp);
    float ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) ds[r] = __expf(s[r] - lse_q) * (dp[r] - d_q);
    bf16x8 dsb[2];
    acc_to_b<F16>(ds, dsb);
    acc = mfma_32x32x16<F16>(ktf[0], dsb[0], acc);          // dQ^T[d_qk][query]
    acc = mfma_32x32x16<F16>(ktf[1], dsb[1], acc);
  }
  store_feat<F16>(dq + row * g.dk, g.dk, kgrp, acc);
}

// ------------------------------------------------------------------------------------------------
// backward, keys on the lanes: dV[j] = sum_i P[i][j] dO_i,  dK[j] = sum_i dS[i][j] q_i
// qt = Q^T [n][dk][len], dot = dO^T [n][dv][len]
// ------------------------------------------------------------------------------------------------
template <int DVB, bool F16>
__global__ __launch_bounds__(256) void flash_bwd_kv_kernel(const bf16* __restrict__ q, const bf16* __restrict__ k,
                                                           const bf16* __restrict__ v, const bf16* __restrict__ qt,
                                                           const bf16* __restrict__ d_o, const bf16* __restrict__ dot,
                                                           const float* __restrict__ lse, const float* __restrict__ dvec,
                                                           bf16* __restrict__ dk_out, bf16* __restrict__ dv_out,
                                                           const FlashGeom g) {
  constexpr int KT = DVB * 2;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, kgrp = lane >> 5, l31 = lane & 31;
  const int img = blockIdx.y;
  const int key0 = (blockIdx.x * 4 + wid) * 32;
  const size_t krow = (size_t)img * g.len + key0 + l31;
  const bf16* qi = q + (size_t)img * g.len * g.dk;
  const bf16* doi = d_o + (size_t)img * g.len * g.dv;
  const bf16* qti = qt + (size_t)img * g.dk * g.len;
  const bf16* doti = dot + (size_t)img * g.dv * g.len;
  const float* lsei = lse + (size_t)img * g.len;
  const float* dvi = dvec + (size_t)img * g.len;
  const bf16x8 kfb = feat_frag(k + (size_t)img * g.len * g.dk, key0 + l31, g.dk, kgrp);      // B of S = Q K^T
  bf16x8 vfb[KT];                                                                           // B of dP = dO V^T
#pragma unroll
  for (int t = 0; t < KT; ++t) vfb[t] = load_frag(v + krow * g.dv + 16 * t + 8 * kgrp);
  f32x16 zero, acck, accv[DVB];
#pragma unroll
  for (int r = 0; r < 16; ++r) zero[r] = acck[r] = 0.f;
#pragma unroll
  for (int d = 0; d < DVB; ++d) accv[d] = zero;
  for (int q0 = 0; q0 < g.len; q0 += 32) {
    const bf16x8 qfa = feat_frag(qi, q0 + l31, g.dk, kgrp);      // A of S: rows = queries
    bf16x8 dofa[KT], dotf[DVB][2], qtf[2];
#pragma unroll
    for (int t = 0; t < KT; ++t) dofa[t] = load_frag(doi + (size_t)(q0 + l31) * g.dv + 16 * t + 8 * kgrp);
#pragma unroll
    for (int d = 0; d < DVB; ++d)
#pragma unroll
      for (int st = 0; st < 2; ++st) dotf[d][st] = load_frag(doti + (size_t)(32 * d + l31) * g.len + q0 + 16 * kgrp + 8 * st);
#pragma unroll
    for (int st = 0; st < 2; ++st)
      qtf[st] = l31 < g.dk ? load_frag(qti + (size_t)l31 * g.len + q0 + 16 * kgrp + 8 * st) : zero_frag();
    float lr[16], dr[16];      // per accumulator ROW = per query of the block: the same for all lanes of a half-wave
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qi_ = q0 + 8 * (r >> 2) + 4 * kgrp + (r & 3);
      lr[r] = lsei[qi_];
      dr[r] = dvi[qi_];
    }
    const f32x16 s = mfma_32x32x16<F16>(qfa, kfb, zero);      // S[query][key]
    f32x16 dp = zero;
#pragma unroll
    for (int t = 0; t < KT; ++t) dp = mfma_32x32x16<F16>(dofa[t], vfb[t], dp);      // dP[query][key]
    float p[16], ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      p[r] = __expf(s[r] - lr[r]);
      ds[r] = p[r] * (dp[r] - dr[r]);
    }
    bf16x8 pb[2], dsb[2];
    acc_to_b<F16>(p, pb);
    acc_to_b<F16>(ds, dsb);
#pragma unroll
    for (int d = 0; d < DVB; ++d) {
      accv[d] = mfma_32x32x16<F16>(dotf[d][0], pb[0], accv[d]);      // dV^T[d_v][key]
      accv[d] = mfma_32x32x16<F16>(dotf[d][1], pb[1], accv[d]);
    }
    acck = mfma_32x32x16<F16>(qtf[0], dsb[0], acck);                   // dK^T[d_qk][key]
    acck = mfma_32x32x16<F16>(qtf[1], dsb[1], acck);
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

extern "C" {

int tg_transpose16(const void* src, void* dst, int batch, int rows, int cols, void* stream) {
  TG_CHECK(src && dst && batch > 0 && rows > 0 && cols > 0, TG_EINVAL, "tg_transpose16: bad arguments");
  hipLaunchKernelGGL(transpose16_kernel, dim3((cols + 31) / 32, (rows + 31) / 32, batch), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)src, (unsigned short*)dst, rows, cols);
  TG_LAUNCH_CHECK("tg_transpose16");
  return TG_OK;
}

int tg_flash_attention_supported(int len, int dk, int dv) {
  return (len % 128 == 0 && (dk == 8 || dk == 16) && (dv == 64 || dv == 128 || dv == 256)) ? 1 : 0;
}

int tg_flash_attention_fwd(const void* q, const void* k, const void* v_t, void* o, float* lse, int n, int len, int dk, int dv,
                           int dtype, void* stream) {
  TG_CHECK(q && k && v_t && o && lse && n > 0, TG_EINVAL, "tg_flash_attention_fwd: bad arguments");
  TG_CHECK(tg_flash_attention_supported(len, dk, dv), TG_ENOSUP,
           "tg_flash_attention_fwd: len %% 128 == 0, d_qk in {8, 16}, d_v in {64, 128, 256} (got %d, %d, %d)", len, dk, dv);
  TG_CHECK(dtype == TG_BF16 || dtype == TG_F16, TG_ENOSUP, "tg_flash_attention_fwd: 16-bit storage only");
  FlashGeom g;
  g.n = n; g.len = len; g.dk = dk; g.dv = dv;
  const dim3 grid(len / 128, n);
  hipStream_t s = (hipStream_t)stream;
#define TG_FL(DVB_)                                                                                                    \
  do {                                                                                                                 \
    if (dtype == TG_F16)                                                                                               \
      hipLaunchKernelGGL((flash_fwd_kernel<DVB_, true>), grid, dim3(256), 0, s, (const bf16*)q, (const bf16*)k,         \
                         (const bf16*)v_t, (bf16*)o, lse, g);                                                           \
    else                                                                                                               \
      hipLaunchKernelGGL((flash_fwd_kernel<DVB_, false>), grid, dim3(256), 0, s, (const bf16*)q, (const bf16*)k,        \
                         (const bf16*)v_t, (bf16*)o, lse, g);                                                           \
  } while (0)
  if (dv == 64) TG_FL(2);
  else if (dv == 128) TG_FL(4);
  else TG_FL(8);
#undef TG_FL
  TG_LAUNCH_CHECK("tg_flash_attention_fwd");
  return TG_OK;
}

int tg_flash_attention_bwd(const void* q, const void* k, const void* v, const void* q_t, const void* k_t, const void* d_o,
                           const void* d_o_t, const void* o, const float* lse, float* dvec, void* dq, void* dk_out, void* dv_out,
                           int n, int len, int dk, int dv, int dtype, void* stream) {
  TG_CHECK(q && k && v && q_t && k_t && d_o && d_o_t && o && lse && dvec && dq && dk_out && dv_out && n > 0, TG_EINVAL,
           "tg_flash_attention_bwd: bad arguments");
  TG_CHECK(tg_flash_attention_supported(len, dk, dv) && dv <= 128, TG_ENOSUP,
           "tg_flash_attention_bwd: len %% 128 == 0, d_qk in {8, 16}, d_v in {64, 128} (got %d, %d, %d)", len, dk, dv);
  TG_CHECK(dtype == TG_BF16 || dtype == TG_F16, TG_ENOSUP, "tg_flash_attention_bwd: 16-bit storage only");
  FlashGeom g;
  g.n = n; g.len = len; g.dk = dk; g.dv = dv;
  hipStream_t s = (hipStream_t)stream;
  const int64_t rows = (int64_t)n * len;
  if (dtype == TG_F16)
    hipLaunchKernelGGL(flash_rowdot_kernel<f16>, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, (const f16*)d_o,
                       (const f16*)o, dvec, rows, dv);
  else
    hipLaunchKernelGGL(flash_rowdot_kernel<bf16>, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, (const bf16*)d_o,
                       (const bf16*)o, dvec, rows, dv);
  const dim3 grid(len / 128, n);
#define TG_FLB(DVB_, F16_)                                                                                              \
  do {                                                                                                                  \
    hipLaunchKernelGGL((flash_bwd_q_kernel<DVB_, F16_>), grid, dim3(256), 0, s, (const bf16*)q, (const bf16*)k,          \
                       (const bf16*)v, (const bf16*)k_t, (const bf16*)d_o, lse, dvec, (bf16*)dq, g);                     \
    hipLaunchKernelGGL((flash_bwd_kv_kernel<DVB_, F16_>), grid, dim3(256), 0, s, (const bf16*)q, (const bf16*)k,         \
                       (const bf16*)v, (const bf16*)q_t, (const bf16*)d_o, (const bf16*)d_o_t, lse, dvec, (bf16*)dk_out, \
                       (bf16*)dv_out, g);                                                                                \
  } while (0)
  if (dv == 64 && dtype == TG_F16) TG_FLB(2, true);
  else if (dv == 64) TG_FLB(2, false);
  else if (dtype == TG_F16) TG_FLB(4, true);
  else TG_FLB(4, false);
#undef TG_FLB
  TG_LAUNCH_CHECK("tg_flash_attention_bwd");
  return TG_OK;
}

}  // extern "C"
