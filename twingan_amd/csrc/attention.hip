// SAGAN self-attention (libs/self_attention.py:24-70) as gfx950 kernels behind the C ABI:
//     f, g = tanh(conv1x1(x))  [N, c/8],  h = conv1x1(x)  [N, c]        (the 1x1 convs are tg_conv2d_* / tg_pointwise_*)
//     s = f g^T  [N, N],  beta = softmax(s, axis=-1),  o = beta h,  y = sa_gamma * o + x           (N = H*W per image)
// built from three primitives, each of which is its own derivative family, so that the layer is differentiable TWICE
// on these kernels (the discriminators sit under the WGAN-GP gradient penalty):
//   * tg_batched_gemm      C[b] = alpha * op(A[b]) op(B[b]) (+ C[b])   -- bf16: v_mfma_f32_32x32x16_bf16, 128 x 64 x 32
//                          workgroup tiles staged through LDS; an operand whose reduction axis is contiguous in memory is
//                          read with ds_read_b128, the other kind with the LDS transpose read ds_read_b64_tr_b16 (row
//                          strides 320 / 192 B keep a half-wave's 4 x 32-byte segments on distinct banks); fp32: one
//                          thread per output element (the exact-parity path, small maps)
//   * tg_softmax_rows_{fwd,bwd,bwd_bwd}   row softmax, its backward dS = P (dP - sum(dP P)) and that map's gradient in P
//   * tg_tanh_{fwd,bwd}, tg_mul3, tg_dot, tg_scale_dev   the pointwise pieces and the sa_gamma scale / reduction
// Replaces tf.matmul x2, tf.nn.softmax, tf.nn.tanh and the gamma * o + layer arithmetic of libs/self_attention.py:57-69.
#include "tg_common.h"

namespace {

struct BgGeom {
  int m, n, k;
  int lda, ldb, ldc;
  long long sa, sb, sc;      // batch strides in elements
  float alpha;
  int accumulate, c_f32, vec;
};

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

__device__ __forceinline__ bf16x8 lds_tr8(const unsigned char* p, int second) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + second));
  s16x8 v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
  v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(bf16x8, v);
}

// one 16-byte vector (8 elements along the contiguous axis) of a [rows][cols] global matrix, zero outside
__device__ __forceinline__ bf16x8 gload8(const bf16* base, int row, int col, int rows, int cols, int ld, bool vec) {
  bf16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (bf16)0.f;
  if (row >= rows || col >= cols) return v;
  const bf16* p = base + (size_t)row * ld + col;
  if (vec) return *reinterpret_cast<const bf16x8*>(p);      // cols % 8 == 0: the vector is entirely inside
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (col + j < cols) v[j] = p[j];
  return v;
}

constexpr int BM = 128, BN = 64, BK = 32;
constexpr int A_KC_STRIDE = 80, B_KC_STRIDE = 80;        // bytes per row of a K-contiguous tile [rows][32 k] (+16 pad)
constexpr int A_KS_STRIDE = 320, B_KS_STRIDE = 192;      // bytes per k row of a K-strided tile [32 k][128 | 64 cols] (+64 pad)

// AKC / BKC: op(A)'s / op(B)'s reduction axis is contiguous in memory (A stored [m][k] / B stored [n][k])
template <bool AKC, bool BKC, bool F16 = false>
__global__ __launch_bounds__(256) void bgemm_mfma_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B,
                                                         void* __restrict__ C, const BgGeom g) {
  __shared__ __attribute__((aligned(16))) unsigned char sA[AKC ? BM * A_KC_STRIDE : BK * A_KS_STRIDE];
  __shared__ __attribute__((aligned(16))) unsigned char sB[BKC ? BN * B_KC_STRIDE : BK * B_KS_STRIDE];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const bf16* Ab = A + (size_t)blockIdx.z * g.sa;
  const bf16* Bb = B + (size_t)blockIdx.z * g.sb;
  const bool vec = g.vec != 0;

  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;

  bf16x8 ra[2], rb;
  auto load_tiles = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int v = tid + s * 256;
      if constexpr (AKC) ra[s] = gload8(Ab, m0 + (v >> 2), k0 + (v & 3) * 8, g.m, g.k, g.lda, vec);
      else ra[s] = gload8(Ab, k0 + (v >> 4), m0 + (v & 15) * 8, g.k, g.m, g.lda, vec);
    }
    if constexpr (BKC) rb = gload8(Bb, n0 + (tid >> 2), k0 + (tid & 3) * 8, g.n, g.k, g.ldb, vec);
    else rb = gload8(Bb, k0 + (tid >> 3), n0 + (tid & 7) * 8, g.k, g.n, g.ldb, vec);
  };
  auto store_tiles = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int v = tid + s * 256;
      if constexpr (AKC) *reinterpret_cast<bf16x8*>(sA + (v >> 2) * A_KC_STRIDE + (v & 3) * 16) = ra[s];
      else *reinterpret_cast<bf16x8*>(sA + (v >> 4) * A_KS_STRIDE + (v & 15) * 16) = ra[s];
    }
    if constexpr (BKC) *reinterpret_cast<bf16x8*>(sB + (tid >> 2) * B_KC_STRIDE + (tid & 3) * 16) = rb;
    else *reinterpret_cast<bf16x8*>(sB + (tid >> 3) * B_KS_STRIDE + (tid & 7) * 16) = rb;
  };

  // fragment addresses: row / column index i = lane & 31, k group kg = lane >> 5 (8 consecutive k);
  // transpose reads: 16-lane group q = lane >> 4 -> column half q & 1, lane s = lane & 15 -> k row +(s >> 2), columns 4 (s & 3)
  const int i32 = lane & 31, kg = lane >> 5, q = lane >> 4, s16 = lane & 15;
  const int a_kc = (wid * 32 + i32) * A_KC_STRIDE + kg * 16;
  const int a_ks = (kg * 8 + (s16 >> 2)) * A_KS_STRIDE + (wid * 32 + (q & 1) * 16 + (s16 & 3) * 4) * 2;
  const int b_kc = i32 * B_KC_STRIDE + kg * 16;
  const int b_ks = (kg * 8 + (s16 >> 2)) * B_KS_STRIDE + ((q & 1) * 16 + (s16 & 3) * 4) * 2;

  load_tiles(0);
  for (int k0 = 0; k0 < g.k; k0 += BK) {
    __syncthreads();      // the previous tile's fragment reads are done
    store_tiles();
    __syncthreads();
    if (k0 + BK < g.k) load_tiles(k0 + BK);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af;
      if constexpr (AKC) af = *reinterpret_cast<const bf16x8*>(sA + a_kc + ks * 32);
      else af = lds_tr8(sA + a_ks + ks * 16 * A_KS_STRIDE, 4 * A_KS_STRIDE);
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        bf16x8 bfr;
        if constexpr (BKC) bfr = *reinterpret_cast<const bf16x8*>(sB + b_kc + nb * 32 * B_KC_STRIDE + ks * 32);
        else bfr = lds_tr8(sB + b_ks + nb * 64 + ks * 16 * B_KS_STRIDE, 4 * B_KS_STRIDE);
        acc[nb] = mfma_32x32x16<F16>(af, bfr, acc[nb]);
      }
    }
  }

  // acc[nb][r]: row m0 + 32 wid + (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column n0 + 32 nb + (lane & 31)
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int col = n0 + nb * 32 + i32;
    if (col >= g.n) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wid * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
      if (row >= g.m) continue;
      const size_t off = (size_t)blockIdx.z * g.sc + (size_t)row * g.ldc + col;
      float v = g.alpha * acc[nb][r];
      if (g.c_f32) {
        float* c = reinterpret_cast<float*>(C) + off;
        *c = g.accumulate ? *c + v : v;
      } else {
        if constexpr (F16) {
          f16* c = reinterpret_cast<f16*>(C) + off;
          *c = (f16)(g.accumulate ? (float)*c + v : v);
        } else {
          bf16* c = reinterpret_cast<bf16*>(C) + off;
          *c = (bf16)(g.accumulate ? (float)*c + v : v);
        }
      }
    }
  }
}

// any dtype, any shape: one thread per output element (fp32 exact-parity path)
template <typename T>
__global__ void bgemm_simple_kernel(const T* __restrict__ A, const T* __restrict__ B, void* __restrict__ C, const BgGeom g,
                                    int ta, int tb) {
  const long long total = (long long)g.m * g.n;
  const T* Ab = A + (size_t)blockIdx.y * g.sa;
  const T* Bb = B + (size_t)blockIdx.y * g.sb;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int row = (int)(i / g.n), col = (int)(i - (long long)row * g.n);
    float acc = 0.f;
    for (int k = 0; k < g.k; ++k) {
      const float a = ld(Ab + (ta ? (size_t)k * g.lda + row : (size_t)row * g.lda + k));
      const float b = ld(Bb + (tb ? (size_t)col * g.ldb + k : (size_t)k * g.ldb + col));
      acc = fmaf(a, b, acc);
    }
    const size_t off = (size_t)blockIdx.y * g.sc + (size_t)row * g.ldc + col;
    const float v = g.alpha * acc;
    if (g.c_f32) {
      float* c = reinterpret_cast<float*>(C) + off;
      *c = g.accumulate ? *c + v : v;
    } else {
      T* c = reinterpret_cast<T*>(C) + off;
      st(c, g.accumulate ? ld(c) + v : v);
    }
  }
}

// ---- row softmax -------------------------------------------------------------------------------------------------
// block = one row of `cols` elements
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_fwd_kernel(const T* __restrict__ s, T* __restrict__ p, int cols) {
  __shared__ float red[4];
  const T* row = s + (size_t)blockIdx.x * cols;
  T* out = p + (size_t)blockIdx.x * cols;
  float mx = -3.0e38f;
  for (int c = threadIdx.x; c < cols; c += 256) mx = fmaxf(mx, ld(row + c));
  // block max through the sum helper's scratch
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int c = threadIdx.x; c < cols; c += 256) sum += __expf(ld(row + c) - mx);
  sum = block_sum(sum, red);
  const float inv = 1.f / sum;
  for (int c = threadIdx.x; c < cols; c += 256) st(out + c, __expf(ld(row + c) - mx) * inv);
}

// ds = p * (dp - sum_j dp_j p_j)
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const T* __restrict__ p, const T* __restrict__ dp,
                                                               T* __restrict__ ds, int cols) {
  __shared__ float red[4];
  const size_t base = (size_t)blockIdx.x * cols;
  float t = 0.f;
  for (int c = threadIdx.x; c < cols; c += 256) t = fmaf(ld(dp + base + c), ld(p + base + c), t);
  t = block_sum(t, red);
  for (int c = threadIdx.x; c < cols; c += 256) st(ds + base + c, ld(p + base + c) * (ld(dp + base + c) - t));
}

// gradient of L = sum_j v_j ds_j (ds as above) with respect to p:  v_j (dp_j - t) - dp_j u,  t = sum dp p, u = sum v p
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_bwd_bwd_kernel(const T* __restrict__ p, const T* __restrict__ dp,
                                                                   const T* __restrict__ v, T* __restrict__ gp, int cols) {
  __shared__ float red[4];
  const size_t base = (size_t)blockIdx.x * cols;
  float t = 0.f, u = 0.f;
  for (int c = threadIdx.x; c < cols; c += 256) {
    const float pc = ld(p + base + c);
    t = fmaf(ld(dp + base + c), pc, t);
    u = fmaf(ld(v + base + c), pc, u);
  }
  t = block_sum(t, red);
  u = block_sum(u, red);
  for (int c = threadIdx.x; c < cols; c += 256) {
    const float d = ld(dp + base + c);
    st(gp + base + c, ld(v + base + c) * (d - t) - d * u);
  }
}

// ---- pointwise pieces ----------------------------------------------------------------------------------------------
template <typename T>
__global__ void tanh_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    st(y + i, tanhf(ld(x + i)));
}
template <typename T>      // gx = g * (1 - y^2)
__global__ void tanh_bwd_kernel(const T* __restrict__ g, const T* __restrict__ y, T* __restrict__ gx, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float yy = ld(y + i);
    st(gx + i, ld(g + i) * (1.f - yy * yy));
  }
}
template <typename T>      // out = scale * a * b * c
__global__ void mul3_kernel(const T* __restrict__ a, const T* __restrict__ b, const T* __restrict__ c, T* __restrict__ out,
                            float scale, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    st(out + i, scale * ld(a + i) * ld(b + i) * (c ? ld(c + i) : 1.f));
}
template <typename T>      // out = x * s[0]  (s: device fp32 scalar)
__global__ void scale_dev_kernel(const T* __restrict__ x, const float* __restrict__ s, T* __restrict__ out, int64_t n) {
  const float k = s[0];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    st(out + i, ld(x + i) * k);
}
// part[block] = sum over the block's range of a * b (fixed order); dot_final sums the partials
template <typename T>
__global__ __launch_bounds__(256) void dot_partial_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                          float* __restrict__ part, int64_t n, int64_t per) {
  __shared__ float red[4];
  const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  float t = 0.f;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) t = fmaf(ld(a + i), ld(b + i), t);
  t = block_sum(t, red);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}
__global__ __launch_bounds__(256) void dot_final_kernel(const float* __restrict__ part, int nparts, float* __restrict__ out) {
  __shared__ float red[4];
  float t = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) t += part[i];
  t = block_sum(t, red);
  if (threadIdx.x == 0) out[0] = t;
}

}  // namespace

extern "C" {

int tg_batched_gemm(const void* a, const void* b, void* c, int batch, int m, int n, int k, int ta, int tb, int lda, int ldb,
                    int ldc, int64_t stride_a, int64_t stride_b, int64_t stride_c, float alpha, int accumulate, int dtype,
                    int c_is_f32, void* stream) {
  TG_CHECK(a && b && c && batch > 0 && m > 0 && n > 0 && k > 0, TG_EINVAL, "tg_batched_gemm: bad arguments");
  TG_CHECK(dtype == TG_F32 || dtype == TG_BF16 || dtype == TG_F16, TG_EINVAL, "tg_batched_gemm: dtype %d", dtype);
  BgGeom g;
  g.m = m; g.n = n; g.k = k; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.sa = stride_a; g.sb = stride_b; g.sc = stride_c;
  g.alpha = alpha; g.accumulate = accumulate; g.c_f32 = c_is_f32 || dtype == TG_F32;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == TG_F32) {
    g.vec = 0;
    const dim3 grid(tg_grid_for((int64_t)m * n, 256, 4096), batch);
    hipLaunchKernelGGL(bgemm_simple_kernel<float>, grid, dim3(256), 0, s, (const float*)a, (const float*)b, c, g, ta, tb);
    TG_LAUNCH_CHECK("tg_batched_gemm(f32)");
    return TG_OK;
  }
  // 16-byte vector staging needs every contiguous run to be a multiple of 8 elements at a 16-byte aligned address
  const int ca = ta ? m : k, cb = tb ? k : n;      // contiguous extents of A and B
  g.vec = (ca % 8 == 0 && cb % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && stride_a % 8 == 0 && stride_b % 8 == 0 &&
           tg_aligned16(a) && tg_aligned16(b)) ? 1 : 0;
  const dim3 grid((n + BN - 1) / BN, (m + BM - 1) / BM, batch);
  const bool akc = !ta, bkc = tb != 0;
  tg_note_kernel("bgemm_mfma_kernel<%d,%d>", (int)akc, (int)bkc);
#define TG_BG_LAUNCH(A_, B_)                                                                                            \
  do {                                                                                                                \
    if (dtype == TG_F16)                                                                                              \
      hipLaunchKernelGGL((bgemm_mfma_kernel<A_, B_, true>), grid, dim3(256), 0, s, (const bf16*)a, (const bf16*)b, c, g); \
    else                                                                                                              \
      hipLaunchKernelGGL((bgemm_mfma_kernel<A_, B_>), grid, dim3(256), 0, s, (const bf16*)a, (const bf16*)b, c, g);     \
  } while (0)
  if (akc && bkc) TG_BG_LAUNCH(true, true);
  else if (akc) TG_BG_LAUNCH(true, false);
  else if (bkc) TG_BG_LAUNCH(false, true);
  else TG_BG_LAUNCH(false, false);
#undef TG_BG_LAUNCH
  TG_LAUNCH_CHECK("tg_batched_gemm");
  return TG_OK;
}

int tg_softmax_rows_fwd(const void* s, void* p, int64_t rows, int cols, int dtype, void* stream) {
  TG_CHECK(s && p && rows > 0 && cols > 0, TG_EINVAL, "tg_softmax_rows_fwd: bad arguments");
  TG_DISPATCH_DTYPE(dtype, "tg_softmax_rows_fwd", {
    hipLaunchKernelGGL(softmax_rows_fwd_kernel<T>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const T*)s, (T*)p, cols);
  });
  TG_LAUNCH_CHECK("tg_softmax_rows_fwd");
  return TG_OK;
}

int tg_softmax_rows_bwd(const void* p, const void* dp, void* ds, int64_t rows, int cols, int dtype, void* stream) {
  TG_CHECK(p && dp && ds && rows > 0 && cols > 0, TG_EINVAL, "tg_softmax_rows_bwd: bad arguments");
  TG_DISPATCH_DTYPE(dtype, "tg_softmax_rows_bwd", {
    hipLaunchKernelGGL(softmax_rows_bwd_kernel<T>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const T*)p,
                       (const T*)dp, (T*)ds, cols);
  });
  TG_LAUNCH_CHECK("tg_softmax_rows_bwd");
  return TG_OK;
}

int tg_softmax_rows_bwd_bwd(const void* p, const void* dp, const void* v, void* gp, int64_t rows, int cols, int dtype,
                            void* stream) {
  TG_CHECK(p && dp && v && gp && rows > 0 && cols > 0, TG_EINVAL, "tg_softmax_rows_bwd_bwd: bad arguments");
  TG_DISPATCH_DTYPE(dtype, "tg_softmax_rows_bwd_bwd", {
    hipLaunchKernelGGL(softmax_rows_bwd_bwd_kernel<T>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const T*)p,
                       (const T*)dp, (const T*)v, (T*)gp, cols);
  });
  TG_LAUNCH_CHECK("tg_softmax_rows_bwd_bwd");
  return TG_OK;
}

int tg_tanh_fwd(const void* x, void* y, int64_t numel, int dtype, void* stream) {
  TG_CHECK(x && y && numel > 0, TG_EINVAL, "tg_tanh_fwd: bad arguments");
  TG_DISPATCH_DTYPE(dtype, "tg_tanh_fwd", {
    hipLaunchKernelGGL(tanh_fwd_kernel<T>, dim3(tg_grid_for(numel, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y, numel);
  });
  TG_LAUNCH_CHECK("tg_tanh_fwd");
  return TG_OK;
}

int tg_tanh_bwd(const void* g, const void* y, void* gx, int64_t numel, int dtype, void* stream) {
  TG_CHECK(g && y && gx && numel > 0, TG_EINVAL, "tg_tanh_bwd: bad arguments");
  TG_DISPATCH_DTYPE(dtype, "tg_tanh_bwd", {
    hipLaunchKernelGGL(tanh_bwd_kernel<T>, dim3(tg_grid_for(numel, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)g,
                       (const T*)y, (T*)gx, numel);
  });
  TG_LAUNCH_CHECK("tg_tanh_bwd");
  return TG_OK;
}

int tg_mul3(const void* a, const void* b, const void* c, void* out, float scale, int64_t numel, int dtype, void* stream) {
  TG_CHECK(a && b && out && numel > 0, TG_EINVAL, "tg_mul3: bad arguments");
  TG_DISPATCH_DTYPE(dtype, "tg_mul3", {
    hipLaunchKernelGGL(mul3_kernel<T>, dim3(tg_grid_for(numel, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)a,
                       (const T*)b, (const T*)c, (T*)out, scale, numel);
  });
  TG_LAUNCH_CHECK("tg_mul3");
  return TG_OK;
}

int tg_scale_dev(const void* x, const float* scalar, void* out, int64_t numel, int dtype, void* stream) {
  TG_CHECK(x && scalar && out && numel > 0, TG_EINVAL, "tg_scale_dev: bad arguments");
  TG_DISPATCH_DTYPE(dtype, "tg_scale_dev", {
    hipLaunchKernelGGL(scale_dev_kernel<T>, dim3(tg_grid_for(numel, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)x,
                       scalar, (T*)out, numel);
  });
  TG_LAUNCH_CHECK("tg_scale_dev");
  return TG_OK;
}

// out[0] = sum a * b; ws: >= 1024 floats of scratch (two-stage, fixed order)
int tg_dot(const void* a, const void* b, float* out, float* ws, int64_t numel, int dtype, void* stream) {
  TG_CHECK(a && b && out && ws && numel > 0, TG_EINVAL, "tg_dot: bad arguments");
  int nparts = (int)((numel + 16383) / 16384);
  if (nparts > 1024) nparts = 1024;
  const int64_t per = (numel + nparts - 1) / nparts;
  TG_DISPATCH_DTYPE(dtype, "tg_dot", {
    hipLaunchKernelGGL(dot_partial_kernel<T>, dim3(nparts), dim3(256), 0, (hipStream_t)stream, (const T*)a, (const T*)b, ws,
                       numel, per);
  });
  hipLaunchKernelGGL(dot_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ws, nparts, out);
  TG_LAUNCH_CHECK("tg_dot");
  return TG_OK;
}

}  // extern "C"
