// Instance norm statistics, the fused (instance-norm affine + LeakyReLU + pixel-norm) forward and
// backward, and per-channel sums (BiasAddGrad).  NHWC, fp32 or bf16 storage, fp32 math.
//
// Reference: libs/instance_norm.py:131-135 (tf.nn.moments over H,W + tf.nn.batch_normalization),
// util_misc.py:68-86 (LeakyReLU), nets/pggan_utils.py:330-331 (pixel norm); layer order
// conv -> norm -> act -> pixel-norm (nets/pggan.py:78-81).
//
// Thread mapping everywhere: one thread owns one V-wide channel vector of one pixel; the C/V
// threads of a pixel are adjacent lanes of one wave, so the per-pixel channel reduction of pixel
// norm is a butterfly shuffle inside the wave.
#include "tg_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

#ifndef TG_NORM_BWD_U
#define TG_NORM_BWD_U 2      // A/B builds: -DTG_NORM_BWD_U=...
#endif
#define NF_LRELU 1
#define NF_PIXNORM 2
#define NF_ZEROSHIFT 8    // the partial sums were taken without a shift (conv epilogue, tg_conv2d_fwd_stats): K = 0
#define NF_NOSTATS 4      // mean / rstd are constants (no normaliser: pixel norm only): the backward drops the statistic terms

template <typename T, int V>
struct VecIO {
  __device__ static __forceinline__ void load(const T* p, float* o) {
    if constexpr (V == 1) {
      o[0] = ld(p);
    } else {
      Vec16<T> v = ldv(p);
#pragma unroll
      for (int j = 0; j < V; ++j) o[j] = v.get(j);
    }
  }
  __device__ static __forceinline__ void store(T* p, const float* o) {
    if constexpr (V == 1) {
      st(p, o[0]);
    } else {
      Vec16<T> v;
#pragma unroll
      for (int j = 0; j < V; ++j) v.set(j, o[j]);
      stv(p, v);
    }
  }
  // the full-resolution output streams: wave-uniform base + 32-bit element offset (TG_STORE_AUX picks the cache policy)
  __device__ static __forceinline__ void store_stream(T* base, unsigned off, const float* o) {
    if constexpr (V == 1) {
      st(base + off, o[0]);
    } else {
      Vec16<T> v;
#pragma unroll
      for (int j = 0; j < V; ++j) v.set(j, o[j]);
      stv_stream(base, off, v);
    }
  }
};

// Incoming layer-output gradient of pixel (n, p) = gz[n,p] (may be NULL) + scale * gzp[n, y/2, x/2] (may be NULL):
// the 2x2 average pool that follows a layer is folded into the layer's backward instead of materialising
// the upsampled pooled gradient (and the sum with the UNet-skip gradient) in HBM.
template <typename T, int V>
__device__ __forceinline__ void load_grad(const T* gz, const T* gzp, int n, int p, int hw, int wdim, int c, int v,
                                          float pool_scale, float* g) {
  if (gz) {
    VecIO<T, V>::load(gz + ((int64_t)n * hw + p) * c + v * V, g);
  } else {
#pragma unroll
    for (int j = 0; j < V; ++j) g[j] = 0.f;
  }
  if (gzp) {
    // wdim / hw are powers of two for every progressive stage: shifts instead of runtime divisions
    const bool p2 = (wdim & (wdim - 1)) == 0;
    const int yy = p2 ? (p >> (31 - __builtin_clz(wdim))) : p / wdim, xx = p - yy * wdim;
    const int hp = p2 ? ((hw >> (31 - __builtin_clz(wdim))) >> 1) : hw / wdim / 2;
    const int64_t pp = ((int64_t)n * hp + (yy >> 1)) * (wdim >> 1) + (xx >> 1);
    float q[V];
    VecIO<T, V>::load(gzp + pp * c + v * V, q);
#pragma unroll
    for (int j = 0; j < V; ++j) g[j] = fmaf(pool_scale, q[j], g[j]);
  }
}

// the same with the image's base pointers (gz_n = gz + n * hw * c, gzp_n = gzp + n * (hw / 4) * c; either may be NULL)
// and 32-bit element offsets: the hot loops' form (an image has < 2^31 elements)
template <typename T, int V>
__device__ __forceinline__ void load_grad_img(const T* gz_n, const T* gzp_n, unsigned p, int wshift, unsigned wdim,
                                              unsigned c, unsigned v, float pool_scale, float* g) {
  if (gz_n) {
    VecIO<T, V>::load(gz_n + (p * c + v * V), g);
  } else {
#pragma unroll
    for (int j = 0; j < V; ++j) g[j] = 0.f;
  }
  if (gzp_n) {
    const unsigned yy = wshift >= 0 ? (p >> wshift) : p / wdim, xx = p - yy * wdim;
    float q[V];
    VecIO<T, V>::load(gzp_n + (((yy >> 1) * (wdim >> 1) + (xx >> 1)) * c + v * V), q);
#pragma unroll
    for (int j = 0; j < V; ++j) g[j] = fmaf(pool_scale, q[j], g[j]);
  }
}

// group-wide (g lanes, power of two <= 64) butterfly sum
__device__ __forceinline__ float group_sum(float v, int g) {
  for (int o = g >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// sh[base + v*V + j] = sum over the block's threads that own channel vector v (v = tid % cv) of a[j], in a FIXED
// order: threads of a wave are summed with an xor butterfly over the pixel-lane bits, the waves' results go to
// per-wave LDS slots and are added in wave order (no float atomics: neither the statistics of the forward pass nor
// anything else computed here depends on which wave arrives first).  Channel counts that are not a power of two
// (<= 64 vectors) take per-thread slots summed in thread order.  Block-collective: contains barriers; blockDim = 256.
template <int V>
__device__ __forceinline__ void wave_channel_accumulate(float (&a)[V], float* sh, int base, int cv, int v, bool active) {
  __shared__ float scratch[256 * V];
  const int tid = threadIdx.x;
  if (!active) {
#pragma unroll
    for (int j = 0; j < V; ++j) a[j] = 0.f;
  }
  if (cv <= 64 && (cv & (cv - 1)) == 0) {
    for (int o = cv; o < 64; o <<= 1) {
#pragma unroll
      for (int j = 0; j < V; ++j) a[j] += __shfl_xor(a[j], o, 64);
    }
    const int wave = tid >> 6, lane = tid & 63;      // lane < cv: lane == v
    if (lane < cv) {
#pragma unroll
      for (int j = 0; j < V; ++j) scratch[(wave * cv + lane) * V + j] = a[j];
    }
    __syncthreads();
    const int nw = blockDim.x >> 6, span = cv * V;
    for (int i = tid; i < span; i += blockDim.x) {
      float t = 0.f;
      for (int w = 0; w < nw; ++w) t += scratch[w * span + i];
      sh[base + i] = t;
    }
  } else {
#pragma unroll
    for (int j = 0; j < V; ++j) scratch[tid * V + j] = a[j];
    __syncthreads();
    const int lanes = blockDim.x / cv;               // thread (pl, v) = tid pl * cv + v
    for (int i = tid; i < cv * V; i += blockDim.x) {
      const int vv = i / V, j = i - vv * V;
      float t = 0.f;
      for (int pl = 0; pl < lanes; ++pl) t += scratch[(pl * cv + vv) * V + j];
      sh[base + i] = t;
    }
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// statistics: shifted sums  S1 = sum(y - K), S2 = sum((y-K)^2) with K = y[n,0,0,c]  (stable in fp32)
// accumulated into mean[] / rstd[] (pre-zeroed), finalised in place.
// grid = (chunks, n).  block = 256 threads = (256/cv pixel lanes) x (cv vectors)
// ------------------------------------------------------------------------------------------------
// PART: write this block's sums to part[n][chunk][2][c] (s1 = part) instead of atomically adding into s1 / s2
template <typename T, int V, bool PART = false>
__global__ void in_stats_partial(const T* __restrict__ y, float* __restrict__ s1, float* __restrict__ s2, int hw, int c,
                                 int px_per_block) {
  extern __shared__ float sh[];   // [2][c]
  const int cv = c / V;
  const int lanes = blockDim.x / cv;
  const int v = threadIdx.x % cv, pl = threadIdx.x / cv;
  const int n = blockIdx.y;
  for (int i = threadIdx.x; i < 2 * c; i += blockDim.x) sh[i] = 0.f;
  __syncthreads();
  const T* base = y + (int64_t)n * hw * c;
  float k[V], a1[V], a2[V];
  VecIO<T, V>::load(base + v * V, k);
#pragma unroll
  for (int j = 0; j < V; ++j) a1[j] = a2[j] = 0.f;
  const int p0 = blockIdx.x * px_per_block;
  const int p1 = min(p0 + px_per_block, hw);
  if (pl < lanes) {
    constexpr int U = 4;      // pixels in flight per thread
    for (int pb = p0 + pl; pb < p1; pb += lanes * U) {
      float x[U][V];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int p = min(pb + u * lanes, p1 - 1);
        VecIO<T, V>::load(base + (int64_t)p * c + v * V, x[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (pb + u * lanes < p1) {
#pragma unroll
          for (int j = 0; j < V; ++j) {
            const float d = x[u][j] - k[j];
            a1[j] += d;
            a2[j] = fmaf(d, d, a2[j]);
          }
        }
      }
    }
  }
  wave_channel_accumulate<V>(a1, sh, 0, cv, v, pl < lanes);
  wave_channel_accumulate<V>(a2, sh, c, cv, v, pl < lanes);
  __syncthreads();
  if (PART) {
    float* out = s1 + ((int64_t)n * gridDim.x + blockIdx.x) * 2 * c;
    for (int i = threadIdx.x; i < 2 * c; i += blockDim.x) out[i] = sh[i];
    return;
  }
  for (int i = threadIdx.x; i < c; i += blockDim.x) {
    atomicAdd(s1 + (int64_t)n * c + i, sh[i]);
    atomicAdd(s2 + (int64_t)n * c + i, sh[c + i]);
  }
}

// sh[0..2c) = sum over the chunks of part[n][chunk][0..2c)  (every thread of the block; ends with a barrier)
// Every workgroup of the streaming passes starts with this: `chunks` (16-64) rows per value, and written as a two-term loop
// the compiler issued two loads per trip and waited for them -- 8-32 dependent L2 round trips (~10 us) in front of a kernel
// that streams for 10-30 us on the mid-size maps (the 64 x 64 ... 16 x 16 layers sat 1.5-1.8 x above their byte time,
// profiles/r03_z_shapes_eager_step.json).  Eight rows in flight per trip, eight accumulators combined in a fixed order.
// Measured and not kept (round 4): thin layers (2c <= 128 columns) summing their rows in 256 / 2c row groups that meet in
// LDS -- one round of loads instead of four at 32 rows x 32 columns.  +0.1 % on the step (those prologues sit in front of
// kernels that stream for 60-190 us), and the changed summation order moved the fp32 parity path of an ill-conditioned
// configuration (equalized learning rate: var = m2 - m1^2 under cancellation) from inside to outside its tolerance.
// EXACT (the fp32 parity path): the two-accumulator order (even rows + odd rows) that path has had since round 1.  Two of
// its model-level comparisons with the float64 oracle are sensitive to this order (conditional batch norm on a style
// embedding: generator gradients 1.0e-4 -> 5.5e-3 rel-L2 with the eight-accumulator order; equalized learning rate:
// 1.2e-2 with row groups) -- configurations whose statistics are ill-conditioned in fp32 whatever the order -- so the
// order those fixtures were validated with stays; speed is not that path's concern.
template <bool EXACT>
__device__ __forceinline__ void reduce_partials(const float* __restrict__ part, int n, int chunks, int c, float* sh) {
  const int64_t row = 2 * (int64_t)c;
  if constexpr (EXACT) {
    for (int i = threadIdx.x; i < 2 * c; i += blockDim.x) {
      const float* p = part + (int64_t)n * chunks * row + i;
      float a = 0.f, b = 0.f;
      int k = 0;
      for (; k + 1 < chunks; k += 2) {
        a += p[(int64_t)k * row];
        b += p[(int64_t)(k + 1) * row];
      }
      if (k < chunks) a += p[(int64_t)k * row];
      sh[i] = a + b;
    }
    __syncthreads();
    return;
  }
  for (int i = threadIdx.x; i < 2 * c; i += blockDim.x) {
    const float* p = part + (int64_t)n * chunks * row + i;
    float a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = 0.f;
    int k = 0;
    for (; k + 8 <= chunks; k += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(k + u) * row];
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += v[u];
    }
    if (k < chunks) {      // the tail: clamped addresses, masked adds -- still all requested together
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(k + u < chunks ? k + u : chunks - 1) * row];
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += (k + u < chunks) ? v[u] : 0.f;
    }
    sh[i] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  }
  __syncthreads();
}

template <typename T>
__global__ void in_stats_final(const T* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd, int n, int hw,
                               int c, float eps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * c) return;
  const int in_ = i / c, ch = i - in_ * c;
  const float k = ld(y + (int64_t)in_ * hw * c + ch);
  const float inv = 1.f / (float)hw;
  const float m1 = mean[i] * inv, m2 = rstd[i] * inv;
  float var = m2 - m1 * m1;
  var = var < 0.f ? 0.f : var;
  mean[i] = k + m1;
  rstd[i] = rsqrtf(var + eps);
}

// ------------------------------------------------------------------------------------------------
// forward:  u = (y - mean) * rstd * gamma + beta ; a = lrelu(u) ; z = a * s, s = rsqrt(mean_c a^2 + eps)
// ------------------------------------------------------------------------------------------------
template <typename T, int V>
__global__ void norm_act_fwd_kernel(const T* __restrict__ y, const float* __restrict__ mean,
                                    const float* __restrict__ rstd, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, const float* __restrict__ gamma2,
                                    const float* __restrict__ beta2, int split, int pstride, T* __restrict__ z,
                                    float* __restrict__ pn_scale, int64_t npix, int hw, int c, int flags, float alpha,
                                    float pn_eps) {
  const int cv = c / V;
  const int64_t total = npix * cv;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  // total is padded up to a multiple of the stride step so that all lanes of a pixel group stay converged
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int v = (int)(i % cv);
    const int64_t p = i / cv;
    const int n = (int)(p / hw);
    float x[V];
    VecIO<T, V>::load(y + p * c + v * V, x);
    // per-domain affine parameters (images >= split: 2nd domain), or one parameter row per image (pstride = c)
    const float* ga = pstride ? gamma + (int64_t)n * pstride : (n < split ? gamma : gamma2);
    const float* be = pstride ? beta + (int64_t)n * pstride : (n < split ? beta : beta2);
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const int ch = v * V + j;
      const float r = rstd[n * c + ch] * ga[ch];
      float u = x[j] * r + (be[ch] - mean[n * c + ch] * r);      // tf.nn.batch_normalization form
      if (flags & NF_LRELU) u = lrelu_f(u, alpha);
      x[j] = u;
      ss = fmaf(u, u, ss);
    }
    if (flags & NF_PIXNORM) {
      ss = group_sum(ss, cv);
      const float s = rsqrtf(ss / (float)c + pn_eps);
#pragma unroll
      for (int j = 0; j < V; ++j) x[j] *= s;
      if (pn_scale && v == 0) pn_scale[p] = s;
    }
    VecIO<T, V>::store(z + p * c + v * V, x);
  }
}

// Same forward with the statistics finalised in the prologue from the partial sums of in_stats_partial<PART>
// (no separate finalise launch, no zero fill, no atomics).  grid = (chunks2, n): a block stays inside one image.
// Block (0, n) also writes mean / rstd of its image for the backward and the moving averages.
// POOL: the 4 pixels a thread keeps in flight are one 2x2 block and the thread also writes their average to zp
// [n, h/2, w/2, c] (the tf.nn.avg_pool after an encoder block, nets/pggan.py:466-468) -- px_per_block then counts
// 2x2 blocks and wdim is the row length.
template <typename T, int V, bool POOL = false, int FL = -1>      // FL: see norm_act_bwd1_kernel
__global__ void norm_act_fwd_part_kernel(const T* __restrict__ y, const float* __restrict__ part, int chunks,
                                         float* __restrict__ mean, float* __restrict__ rstd,
                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                         const float* __restrict__ gamma2, const float* __restrict__ beta2, int split,
                                         T* __restrict__ z, T* __restrict__ zp, int wdim, float* __restrict__ pn_scale,
                                         int hw, int c, int flags_rt, float alpha, float in_eps, float pn_eps,
                                         int px_per_block) {
  extern __shared__ float sh[];   // [2][c]: sums, then (scale, shift)
  const int flags = FL < 0 ? flags_rt : ((flags_rt & ~3) | FL);
  const int cv = c / V;
  const int lanes = blockDim.x / cv;
  const int v = threadIdx.x % cv, pl = threadIdx.x / cv;
  const int n = blockIdx.y;
  const float* ga = n < split ? gamma : gamma2;
  const float* be = n < split ? beta : beta2;
  // this thread's gamma / beta values are requested BEFORE the partial sums: behind the prologue's barriers they were a
  // second dependent round trip of every workgroup (the loads do not depend on the statistics)
  float ga_[8], be_[8], k_[8];      // c <= 2048: at most 8 channels per thread; k: the shift of the shifted sums
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int i = threadIdx.x + q * blockDim.x;
    ga_[q] = i < c ? ga[i] : 0.f;
    be_[q] = i < c ? be[i] : 0.f;
    k_[q] = (i < c && !(flags & NF_ZEROSHIFT)) ? ld(y + (int64_t)n * hw * c + i) : 0.f;
  }
  reduce_partials<exact_path<T>()>(part, n, chunks, c, sh);
  const float inv = 1.f / (float)hw;
  float m_[8], r_[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {      // compile-time indices: the per-thread rows stay in registers
    const int i = threadIdx.x + q * blockDim.x;
    if (i < c) {
      const float m1 = sh[i] * inv, m2 = sh[c + i] * inv;
      float var = m2 - m1 * m1;
      var = var < 0.f ? 0.f : var;
      m_[q] = k_[q] + m1;
      r_[q] = rsqrtf(var + in_eps);
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int i = threadIdx.x + q * blockDim.x;
    if (i < c) {
      if (blockIdx.x == 0) {
        mean[n * c + i] = m_[q];
        rstd[n * c + i] = r_[q];
      }
      const float r = r_[q] * ga_[q];
      sh[i] = r;                                  // u = y * scale + shift (tf.nn.batch_normalization form)
      sh[c + i] = be_[q] - m_[q] * r;
    }
  }
  __syncthreads();
  float sc[V], sf[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    sc[j] = sh[v * V + j];
    sf[j] = sh[c + v * V + j];
  }
  constexpr int U = 4;      // pixels in flight per thread
  const int nunits = POOL ? hw / 4 : hw;      // 2x2 blocks or pixels of this image
  const int p0 = blockIdx.x * px_per_block;
  const int p1 = min(p0 + px_per_block, nunits);
  if (pl >= lanes) return;
  const unsigned wq = wdim >> 1;
  const int wqshift = (wq & (wq - 1)) == 0 ? 31 - __builtin_clz(wq) : -1;      // power-of-two maps: shift, not divide
  // the image's base pointers; element offsets inside an image fit 32 bits
  const T* y_n = y + (int64_t)n * hw * c;
  T* z_n = z + (int64_t)n * hw * c;
  T* zp_n = POOL ? zp + (int64_t)n * (hw / 4) * c : nullptr;
  float* pn_n = pn_scale ? pn_scale + (int64_t)n * hw : nullptr;
  for (int pb = p0 + pl; pb < p1; pb += lanes * (POOL ? 1 : U)) {      // pl is uniform within a pixel group: groups stay converged
    float x[U][V];
    unsigned px[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (POOL) {
        const unsigned qy = wqshift >= 0 ? ((unsigned)pb >> wqshift) : (unsigned)pb / wq, qx = pb - qy * wq;
        px[u] = (2 * qy + (u >> 1)) * wdim + 2 * qx + (u & 1);
      } else {
        px[u] = min(pb + u * lanes, p1 - 1);
      }
      VecIO<T, V>::load(y_n + (px[u] * c + v * V), x[u]);
    }
    float pooled[V];
#pragma unroll
    for (int j = 0; j < V; ++j) pooled[j] = 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool live = POOL || pb + u * lanes < p1;
      float ss = 0.f;
#pragma unroll
      for (int j = 0; j < V; ++j) {
        float t = x[u][j] * sc[j] + sf[j];
        if (flags & NF_LRELU) t = lrelu_f(t, alpha);
        x[u][j] = t;
        ss = fmaf(t, t, ss);
      }
      if (flags & NF_PIXNORM) {
        ss = group_sum(ss, cv);
        const float q = rsqrtf(ss / (float)c + pn_eps);
#pragma unroll
        for (int j = 0; j < V; ++j) x[u][j] *= q;
        if (pn_n && v == 0 && live) pn_n[px[u]] = q;
      }
      if (live) VecIO<T, V>::store_stream(z_n, px[u] * c + v * V, x[u]);
      if (POOL) {
#pragma unroll
        for (int j = 0; j < V; ++j) pooled[j] += rnd<T>(x[u][j]);      // the pool reads the stored (rounded) z
      }
    }
    if (POOL) {
#pragma unroll
      for (int j = 0; j < V; ++j) pooled[j] *= 0.25f;
      VecIO<T, V>::store(zp_n + ((unsigned)pb * c + v * V), pooled);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward.  gu = d loss / d u (u = pre-activation) of one pixel from (gz, y): recomputed by BOTH passes instead
// of being written by pass 1 and read back by pass 2 (5 tensor passes instead of 6).
//   in: g = gz, x = y, s = pixel-norm scale;  out: g = gu, yh = yhat
// ------------------------------------------------------------------------------------------------
template <int V>
__device__ __forceinline__ void norm_act_gu(float (&g)[V], float (&x)[V], float s, const float (&mu)[V],
                                            const float (&rs)[V], const float (&ga)[V], const float (&be)[V], int flags,
                                            float alpha, int cv, float inv_c, float (&yh)[V]) {
  float f[V];      // LeakyReLU slope of the element: lrelu(u) = u * f, d lrelu / du = f
  float dot = 0.f;
#pragma unroll
  for (int j = 0; j < V; ++j) {
    yh[j] = (x[j] - mu[j]) * rs[j];
    const float u = yh[j] * ga[j] + be[j];
    f[j] = ((flags & NF_LRELU) && !(u > 0.f)) ? alpha : 1.f;
    x[j] = u * f[j] * s;                   // z
    dot = fmaf(g[j], x[j], dot);           // gz . z
  }
  if (flags & NF_PIXNORM) {
    dot = group_sum(dot, cv) * inv_c;
#pragma unroll
    for (int j = 0; j < V; ++j) g[j] = s * (g[j] - x[j] * dot);     // d/da
  }
  if (flags & NF_LRELU) {
#pragma unroll
    for (int j = 0; j < V; ++j) g[j] *= f[j];
  }
}

// Measured and NOT kept (round 2): the backward as ONE kernel with 3 tensor passes -- an image split over G workgroups
// that keep their rows of (gz, y) in registers (9 VGPRs per 16-byte row pair, 8 rows per thread at 2 workgroups per
// CU) between the reduction and the apply and meet on a per-image counter in global memory (no grid barrier,
// self-resetting counters, bounded wait with a self-sufficient slow path; it passed the parity tests).  With
// agent-scope fences every workgroup wrote back / invalidated its XCD's whole L2 (20x slower than two kernels); with
// relaxed agent-scope atomics only (partials, counter, one flag per waiter so that nobody polls a shared line) the
// rendezvous still costs ~15 us of memory round trips against ~7 us of streaming per workgroup, and two workgroups
// per CU cannot hide it: 172 vs 158 us at 256x256x16 n64, 43 vs 27 us at 16x16x256 n64.  Keeping an image on one XCD
// and meeting through its L2 with workgroup-scope atomics timed out (block -> XCD placement and L2-scope visibility
// are not something a kernel can rely on).  Without a rendezvous (images one workgroup can hold: 4x4 / 8x8) it saves
// 14 launches and nothing measurable (836 vs 841 images/s).  A kernel boundary IS the cheap barrier on this chip.
//
// backward pass 1: partial sums  sums[n][chunk][0..c) = sum gu,  [c..2c) = sum gu * yhat
// FL >= 0: the LeakyReLU / pixel-norm bits of `flags` as a compile-time constant (the 16-bit vector paths: no flag tests
// or branches in the pixel loops); FL < 0: all of `flags` at run time -- the scalar paths and the fp32 exact-parity
// path, whose arithmetic stays the one the parity tests were pinned on (the specialised code contracts its multiply-adds
// differently: last-bit changes that Adam's sign-like first steps turn into a 2e-2 difference of the 4x4 stage's update).
template <typename T, int V, int FL = -1>
__global__ void norm_act_bwd1_kernel(const T* __restrict__ gz, const T* __restrict__ gzp, int wdim,
                                     const T* __restrict__ y, const float* __restrict__ pn_scale,
                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                     const float* __restrict__ gamma2, const float* __restrict__ beta2, int split,
                                     int pstride, float* __restrict__ sums, int hw, int c, int flags_rt, float alpha,
                                     int px_per_block) {
  extern __shared__ float sh[];   // [2][c]
  const int flags = FL < 0 ? flags_rt : ((flags_rt & ~3) | FL);
  const int cv = c / V;
  const int lanes = blockDim.x / cv;
  const int v = threadIdx.x % cv, pl = threadIdx.x / cv;
  const int n = blockIdx.y;
  for (int i = threadIdx.x; i < 2 * c; i += blockDim.x) sh[i] = 0.f;
  __syncthreads();
  float mu[V], rs[V], ga[V], be[V], a1[V], a2[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    const int ch = v * V + j;
    mu[j] = mean[n * c + ch];
    rs[j] = rstd[n * c + ch];
    ga[j] = pstride ? gamma[(int64_t)n * pstride + ch] : (n < split ? gamma : gamma2)[ch];
    be[j] = pstride ? beta[(int64_t)n * pstride + ch] : (n < split ? beta : beta2)[ch];
    a1[j] = a2[j] = 0.f;
  }
  const int p0 = blockIdx.x * px_per_block;
  const int p1 = min(p0 + px_per_block, hw);
  // the image's base pointers; element offsets inside an image fit 32 bits
  const T* gz_n = gz ? gz + (int64_t)n * hw * c : nullptr;
  const T* gzp_n = gzp ? gzp + (int64_t)n * (hw >> 2) * c : nullptr;
  const T* y_n = y + (int64_t)n * hw * c;
  const float* pn_n = pn_scale ? pn_scale + (int64_t)n * hw : nullptr;
  const int wshift = (wdim & (wdim - 1)) == 0 ? 31 - __builtin_clz(wdim) : -1;
  // every lane of a pixel group iterates the same number of times (pl is uniform within a group)
  const float inv_c = 1.f / (float)c;
  if (pl < lanes) {
    constexpr int U = TG_NORM_BWD_U;      // pixels in flight per thread (3 loads each)
    // TAIL: the block's pixel count is not a multiple of lanes * U -- the last trip clamps and masks its dead pixels
    auto sweep = [&](auto tail) __attribute__((always_inline)) {
      constexpr bool TAIL = decltype(tail)::value;
      for (int pb = p0 + pl; pb < p1; pb += lanes * U) {
        float gq[U][V], xq[U][V], sq[U];
#pragma unroll
        for (int q = 0; q < U; ++q) {
          const unsigned p = TAIL ? min(pb + q * lanes, p1 - 1) : pb + q * lanes;
          load_grad_img<T, V>(gz_n, gzp_n, p, wshift, wdim, c, v, 0.25f, gq[q]);
          VecIO<T, V>::load(y_n + (p * c + v * V), xq[q]);
          sq[q] = (flags & NF_PIXNORM) ? pn_n[p] : 1.f;
        }
#pragma unroll
        for (int q = 0; q < U; ++q) {
          const bool live = !TAIL || pb + q * lanes < p1;
          float yh[V];
          norm_act_gu<V>(gq[q], xq[q], sq[q], mu, rs, ga, be, flags, alpha, cv, inv_c, yh);
#pragma unroll
          for (int j = 0; j < V; ++j) {
            const float gr = live ? gq[q][j] : 0.f;
            a1[j] += gr;
            a2[j] = fmaf(gr, yh[j], a2[j]);
          }
        }
      }
    };
    if ((p1 - p0) % (lanes * U) == 0) sweep(std::false_type());
    else sweep(std::true_type());
  }
  wave_channel_accumulate<V>(a1, sh, 0, cv, v, pl < lanes);
  wave_channel_accumulate<V>(a2, sh, c, cv, v, pl < lanes);
  __syncthreads();
  // this block's partial sums: sums[n][chunk][0..c) = S1, [c..2c) = S2 (no zero fill, no atomics)
  float* out = sums + ((int64_t)n * gridDim.x + blockIdx.x) * 2 * c;
  for (int i = threadIdx.x; i < 2 * c; i += blockDim.x) out[i] = sh[i];
}

// backward pass 2: gy = gamma*rstd * (gu - S1/hw - yhat * S2/hw).  grid = (chunks2, n); the prologue sums the
// image's partial S1 / S2; with `sink` block (0, n) adds them into the parameter gradients.
template <typename T, int V, int FL = -1>
__global__ void norm_act_bwd2_part_kernel(const T* __restrict__ gz, const T* __restrict__ gzp, int wdim,
                                          const T* __restrict__ y, const float* __restrict__ pn_scale,
                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                          const float* __restrict__ gamma2, const float* __restrict__ beta2, int split,
                                          int pstride, T* __restrict__ gy, const float* __restrict__ part, int chunks,
                                          float* __restrict__ ggamma, float* __restrict__ gbeta,
                                          float* __restrict__ ggamma2, float* __restrict__ gbeta2, int sink, int hw, int c,
                                          int flags_rt, float alpha, int px_per_block) {
  extern __shared__ float sh[];   // [2][c]
  const int flags = FL < 0 ? flags_rt : ((flags_rt & ~3) | FL);
  const int cv = c / V;
  const int lanes = blockDim.x / cv;
  const int v = threadIdx.x % cv, pl = threadIdx.x / cv;
  const int n = blockIdx.y;
  // the per-channel constants do not depend on the partial sums: requested before them (one dependent round trip less)
  float mu[V], rs[V], ga[V], be[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    const int ch = v * V + j;
    mu[j] = mean[n * c + ch];
    rs[j] = rstd[n * c + ch];
    ga[j] = pstride ? gamma[(int64_t)n * pstride + ch] : (n < split ? gamma : gamma2)[ch];
    be[j] = pstride ? beta[(int64_t)n * pstride + ch] : (n < split ? beta : beta2)[ch];
  }
  reduce_partials<exact_path<T>()>(part, n, chunks, c, sh);
  if (pstride && blockIdx.x == 0) {      // one parameter row per image: its gradient row is this image's sums
    for (int i = threadIdx.x; i < c; i += blockDim.x) {
      if (gbeta) gbeta[(int64_t)n * pstride + i] = sh[i];
      if (ggamma) ggamma[(int64_t)n * pstride + i] = sh[c + i];
    }
  } else if (sink && blockIdx.x == 0) {
    float* gg = n < split ? ggamma : ggamma2;
    float* gb = n < split ? gbeta : gbeta2;
    if (sink == 2) {
      // exact-parity (fp32) path and deterministic mode: the domain's first image sums the partials of all its images
      // in image order and issues ONE add per parameter (launches that feed a sink are stream-ordered) -- the result
      // does not depend on which image's block arrives first
      const int i0 = n < split ? 0 : split, i1 = n < split ? split : (int)gridDim.y;
      if (n == i0) {
        for (int i = threadIdx.x; i < 2 * c; i += blockDim.x) {
          float t = 0.f;
          const int64_t k1 = (int64_t)i1 * chunks;
          int64_t k = (int64_t)i0 * chunks;
          for (; k + 8 <= k1; k += 8) {      // the same order of adds, eight loads in flight
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(k + u) * 2 * c + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) t += v[u];
          }
          for (; k < k1; ++k) t += part[k * 2 * c + i];
          float* dst = i < c ? gb : gg;
          if (dst) atomicAdd(dst + (i < c ? i : i - c), t);
        }
      }
    } else {
      for (int i = threadIdx.x; i < c; i += blockDim.x) {
        if (gb) atomicAdd(gb + i, sh[i]);
        if (gg) atomicAdd(gg + i, sh[c + i]);
      }
    }
  }
  const float inv = 1.f / (float)hw;
  float s1[V], s2[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    const int ch = v * V + j;
    s1[j] = (flags & NF_NOSTATS) ? 0.f : sh[ch] * inv;
    s2[j] = (flags & NF_NOSTATS) ? 0.f : sh[c + ch] * inv;
  }
  const int p0 = blockIdx.x * px_per_block;
  const int p1 = min(p0 + px_per_block, hw);
  if (pl >= lanes) return;
  const T* gz_n = gz ? gz + (int64_t)n * hw * c : nullptr;
  const T* gzp_n = gzp ? gzp + (int64_t)n * (hw >> 2) * c : nullptr;
  const T* y_n = y + (int64_t)n * hw * c;
  T* gy_n = gy + (int64_t)n * hw * c;
  const float* pn_n = pn_scale ? pn_scale + (int64_t)n * hw : nullptr;
  const int wshift = (wdim & (wdim - 1)) == 0 ? 31 - __builtin_clz(wdim) : -1;
  float gr[V];
#pragma unroll
  for (int j = 0; j < V; ++j) gr[j] = ga[j] * rs[j];
  const float inv_c = 1.f / (float)c;
  constexpr int U = TG_NORM_BWD_U;      // pixels in flight per thread (3 loads each)
  auto sweep = [&](auto tail) __attribute__((always_inline)) {
    constexpr bool TAIL = decltype(tail)::value;
    for (int pb = p0 + pl; pb < p1; pb += lanes * U) {
      float gq[U][V], xq[U][V], sq[U];
#pragma unroll
      for (int q = 0; q < U; ++q) {
        const unsigned p = TAIL ? min(pb + q * lanes, p1 - 1) : pb + q * lanes;
        load_grad_img<T, V>(gz_n, gzp_n, p, wshift, wdim, c, v, 0.25f, gq[q]);
        VecIO<T, V>::load(y_n + (p * c + v * V), xq[q]);
        sq[q] = (flags & NF_PIXNORM) ? pn_n[p] : 1.f;
      }
#pragma unroll
      for (int q = 0; q < U; ++q) {
        const unsigned p = pb + q * lanes;
        float yh[V];
        norm_act_gu<V>(gq[q], xq[q], sq[q], mu, rs, ga, be, flags, alpha, cv, inv_c, yh);
#pragma unroll
        for (int j = 0; j < V; ++j) gq[q][j] = gr[j] * (gq[q][j] - s1[j] - yh[j] * s2[j]);
        if (!TAIL || p < (unsigned)p1) VecIO<T, V>::store_stream(gy_n, p * c + v * V, gq[q]);
      }
    }
  };
  if ((p1 - p0) % (lanes * U) == 0) sweep(std::false_type());
  else sweep(std::true_type());
}

// images [i0, i1) -> (ggamma, gbeta); blockIdx.y selects the domain half
__global__ void norm_param_grads(const float* __restrict__ part, int chunks, float* __restrict__ ggamma,
                                 float* __restrict__ gbeta, float* __restrict__ ggamma2, float* __restrict__ gbeta2,
                                 int split, int n, int c, int accumulate) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= c) return;
  const int i0 = blockIdx.y ? split : 0, i1 = blockIdx.y ? n : split;
  float* gg = blockIdx.y ? ggamma2 : ggamma;
  float* gb = blockIdx.y ? gbeta2 : gbeta;
  float sb = 0.f, sg = 0.f;
  for (int64_t k = (int64_t)i0 * chunks; k < (int64_t)i1 * chunks; ++k) {      // part[n][chunk][2][c]
    sb += part[k * 2 * c + ch];
    sg += part[k * 2 * c + c + ch];
  }
  if (gg) {
    if (accumulate) atomicAdd(gg + ch, sg);
    else gg[ch] = sg;
  }
  if (gb) {
    if (accumulate) atomicAdd(gb + ch, sb);
    else gb[ch] = sb;
  }
}

// out[c] += sum_p g[p][c].  PART: out[blockIdx.x][c] = this workgroup's sum instead (no atomics: tg_channel_sum_ordered adds
// the workgroups' rows in workgroup order)
template <typename T, int V, bool PART = false>
__global__ void channel_sum_kernel(const T* __restrict__ g, float* __restrict__ out, int64_t npix, int c) {
  extern __shared__ float sh[];   // [c]
  const int cv = c / V;
  const int lanes = blockDim.x / cv;
  const int v = threadIdx.x % cv, pl = threadIdx.x / cv;
  for (int i = threadIdx.x; i < c; i += blockDim.x) sh[i] = 0.f;
  __syncthreads();
  float a[V];
#pragma unroll
  for (int j = 0; j < V; ++j) a[j] = 0.f;
  if (pl < lanes) {
    for (int64_t p = (int64_t)blockIdx.x * lanes + pl; p < npix; p += (int64_t)gridDim.x * lanes) {
      float x[V];
      VecIO<T, V>::load(g + p * c + v * V, x);
#pragma unroll
      for (int j = 0; j < V; ++j) a[j] += x[j];
    }
  }
  wave_channel_accumulate<V>(a, sh, 0, cv, v, pl < lanes);
  __syncthreads();
  if constexpr (PART) {
    for (int i = threadIdx.x; i < c; i += blockDim.x) out[(size_t)blockIdx.x * c + i] = sh[i];
  } else {
    for (int i = threadIdx.x; i < c; i += blockDim.x) atomicAdd(out + i, sh[i]);
  }
}

// out[i] (+)= scale * sum over rows b = 0 .. nb-1 of part[b][i], in row order (one thread per element: a fixed order)
__global__ void ordered_rows_sum_kernel(const float* __restrict__ part, int nb, int c, float* __restrict__ out, int accumulate,
                                        float scale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c) return;
  float a[4] = {0.f, 0.f, 0.f, 0.f};      // four interleaved chains (loads in flight), combined in a fixed order
  int b = 0;
  for (; b + 3 < nb; b += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] += part[(size_t)(b + u) * c + i];
  }
  for (; b < nb; ++b) a[0] += part[(size_t)b * c + i];
  const float t = ((a[0] + a[1]) + (a[2] + a[3])) * scale;
  out[i] = accumulate ? out[i] + t : t;
}

// gy = gz * (z > 0 ? 1 : alpha)  and  gbias[c] += sum_p gy[p][c]   (LeakyReLU backward fused with BiasAddGrad)
// BITS (V = 8 only): `z` is not the activation but its sign bits, one byte per 8 channels ([npix][c/8], bit j of byte q =
// (z[.., 8q+j] > 0)): what tg_conv2d_fwd_pool_signs left of a block-end conv's output
template <typename T, int V, bool BITS = false>
__global__ void lrelu_bwd_bias_kernel(const T* __restrict__ gz, const T* __restrict__ gzp, int hw, int wdim,
                                      const T* __restrict__ z, T* __restrict__ gy, float* __restrict__ gbias,
                                      int64_t npix, int c, float alpha) {
  static_assert(!BITS || V == 8, "sign bits come one byte per 16-byte vector of a 16-bit type");
  extern __shared__ float sh[];   // [c]
  constexpr int U = 4;            // pixels in flight per thread: a streaming pass needs >= 64 KB outstanding per CU
  const int cv = c / V;
  const int lanes = blockDim.x / cv;
  const int v = threadIdx.x % cv, pl = threadIdx.x / cv;
  for (int i = threadIdx.x; i < c; i += blockDim.x) sh[i] = 0.f;
  __syncthreads();
  float a[V];
#pragma unroll
  for (int j = 0; j < V; ++j) a[j] = 0.f;
  if (pl < lanes) {
    const int64_t stride = (int64_t)gridDim.x * lanes;
    const int hw_shift = ((hw & (hw - 1)) == 0) ? 31 - __builtin_clz(hw) : -1;
    for (int64_t pb = (int64_t)blockIdx.x * lanes + pl; pb < npix; pb += stride * U) {
      float g[U][V], zz[U][V];
      unsigned zb[U];
      // all loads first (addresses clamped instead of predicated: no branches between the loads)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        int64_t p = pb + u * stride;
        p = p < npix ? p : npix - 1;
        if (gzp) {
          const int n = hw_shift >= 0 ? (int)(p >> hw_shift) : (int)(p / hw);
          load_grad<T, V>(gz, gzp, n, (int)(p - (int64_t)n * hw), hw, wdim, c, v, 0.25f, g[u]);
        } else {
          VecIO<T, V>::load(gz + p * c + v * V, g[u]);
        }
        if constexpr (BITS) zb[u] = reinterpret_cast<const unsigned char*>(z)[p * cv + v];
        else VecIO<T, V>::load(z + p * c + v * V, zz[u]);
      }
      if constexpr (BITS) {      // unfold the bits into the +-1 the arithmetic below tests
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int j = 0; j < V; ++j) zz[u][j] = ((zb[u] >> j) & 1u) ? 1.f : -1.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t p = pb + u * stride;
        if (p < npix) {
#pragma unroll
          for (int j = 0; j < V; ++j) {
            g[u][j] = rnd<T>(g[u][j] * (zz[u][j] > 0.f ? 1.f : alpha));
            a[j] += g[u][j];
          }
#if TG_STORE_AUX
          if (npix * c < (1ll << 31)) VecIO<T, V>::store_stream(gy, (unsigned)(p * c + v * V), g[u]);
          else
#endif
          VecIO<T, V>::store(gy + p * c + v * V, g[u]);
        }
      }
    }
  }
  if (!gbias) return;
  wave_channel_accumulate<V>(a, sh, 0, cv, v, pl < lanes);
  __syncthreads();
  for (int i = threadIdx.x; i < c; i += blockDim.x) atomicAdd(gbias + i, sh[i]);
}

// largest power of two <= 64 check for the pixel-norm group reduction
inline bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

template <typename T>
int pick_v(int c) {
  constexpr int V = Vec16<T>::N;
  return (c % V == 0 && c / V <= 256) ? V : 1;
}

}  // namespace

// smallest pixel range worth a workgroup: small images are latency-bound (one dependent load round per trip), so
// they get one trip per workgroup; larger ones amortise the prologue / partial-sum traffic over >= 64 pixels
static inline int min_ppb(int hw) { return hw <= 64 ? 16 : hw <= 256 ? 32 : 64; }

// pixel range of one statistics / reduction block: ~1024 blocks in total
static void norm_chunks(int n, int hw, int* chunks, int* ppb) {
  int ch = (1024 + n - 1) / n;
  int pp = (hw + ch - 1) / ch;
  if (pp < min_ppb(hw)) pp = min_ppb(hw);
  *ppb = pp;
  *chunks = (hw + pp - 1) / pp;
}

extern "C" {

int tg_norm_chunks(int n, int h, int w) {
  int chunks, ppb;
  if (n <= 0 || h <= 0 || w <= 0) return 0;
  norm_chunks(n, h * w, &chunks, &ppb);
  return chunks;
}

int tg_instance_norm_partials(const void* y, float* partials, int n, int h, int w, int c, int dtype, void* stream) {
  TG_CHECK(y && partials && n > 0 && h > 0 && w > 0 && c > 0, TG_EINVAL, "tg_instance_norm_partials: bad arguments");
  TG_CHECK(c <= 256 * 8, TG_ENOSUP, "tg_instance_norm_partials: c=%d too large", c);
  hipStream_t s = (hipStream_t)stream;
  const int hw = h * w;
  int chunks, ppb;
  norm_chunks(n, hw, &chunks, &ppb);
  TG_DISPATCH_DTYPE(dtype, "tg_instance_norm_partials", {
    const int V = pick_v<T>(c);
    TG_CHECK(c / V <= 256, TG_ENOSUP, "tg_instance_norm_partials: c=%d not supported", c);
    const size_t lds = 2 * (size_t)c * sizeof(float);
    if (V == 1)
      hipLaunchKernelGGL((in_stats_partial<T, 1, true>), dim3(chunks, n), dim3(256), lds, s, (const T*)y, partials, nullptr,
                         hw, c, ppb);
    else
      hipLaunchKernelGGL((in_stats_partial<T, Vec16<T>::N, true>), dim3(chunks, n), dim3(256), lds, s, (const T*)y,
                         partials, nullptr, hw, c, ppb);
  });
  TG_LAUNCH_CHECK("tg_instance_norm_partials");
  return TG_OK;
}

// the forward launch of the vector path with the LeakyReLU / pixel-norm bits FL_ fixed at compile time (uses the locals
// of norm_act_fwd_partials_impl)
#define TG_NF_LAUNCH(POOL_, FL_)                                                                                         \
  hipLaunchKernelGGL((norm_act_fwd_part_kernel<T, VN, POOL_, exact_path<T>() ? -1 : FL_>), dim3(chunks2, n), dim3(256), lds, \
                     (hipStream_t)stream, (const T*)y, partials, chunks, mean, rstd, gamma, beta, gamma2, beta2, split,  \
                     (T*)z, (T*)z_pooled, w, pn_scale, hw, c, flags, alpha, in_eps, pn_eps, ppb)

static int norm_act_fwd_partials_impl(const void* y, const float* partials, int part_chunks, float* mean, float* rstd,
                                      const float* gamma, const float* beta, const float* gamma2, const float* beta2,
                                      int split, void* z, void* z_pooled, float* pn_scale, int n, int h, int w, int c,
                                      int flags, float alpha, float in_eps, float pn_eps, int dtype, void* stream) {
  TG_CHECK(!z_pooled || (h % 2 == 0 && w % 2 == 0), TG_EINVAL, "tg_norm_act_fwd_partials: pooled output needs even h, w");
  TG_CHECK(y && partials && mean && rstd && gamma && beta && z && n > 0 && h > 0 && w > 0 && c > 0, TG_EINVAL,
           "tg_norm_act_fwd_partials: bad arguments");
  TG_CHECK(c <= 256 * 8, TG_ENOSUP, "tg_norm_act_fwd_partials: c=%d too large", c);
  if (!gamma2 || !beta2) split = n;
  TG_CHECK(split >= 0 && split <= n, TG_EINVAL, "tg_norm_act_fwd_partials: split %d outside [0, %d]", split, n);
  const int hw = h * w;
  int chunks, ppb_s;
  norm_chunks(n, hw, &chunks, &ppb_s);
  if (part_chunks > 0) chunks = part_chunks;      // partial sums of a producer with its own chunking
  const int units = z_pooled ? hw / 4 : hw;            // the pooled variant walks 2x2 blocks (4 pixels each)
  int chunks2 = (2048 + n - 1) / n;                    // ~2048 blocks for the streaming pass
  int ppb = (units + chunks2 - 1) / chunks2;
  const int floor_ppb = z_pooled ? 16 : min_ppb(hw);
  if (ppb < floor_ppb) ppb = floor_ppb;
  chunks2 = (units + ppb - 1) / ppb;
  const size_t lds = 2 * (size_t)c * sizeof(float);
  TG_DISPATCH_DTYPE(dtype, "tg_norm_act_fwd_partials", {
    constexpr int VN = Vec16<T>::N;
    const bool vec = (c % VN == 0) && pow2(c / VN) && c / VN <= 64;
    if (flags & NF_PIXNORM) {
      TG_CHECK(vec, TG_ENOSUP, "tg_norm_act_fwd_partials: pixel norm needs c (%d) = %d * 2^k <= %d", c, VN, 64 * VN);
      TG_CHECK(pn_scale, TG_EINVAL, "tg_norm_act_fwd_partials: pixel norm needs pn_scale");
    }
    if (vec) {
      switch ((flags & 3) | (z_pooled ? 4 : 0)) {
        case 0: TG_NF_LAUNCH(false, 0); break;
        case 1: TG_NF_LAUNCH(false, 1); break;
        case 2: TG_NF_LAUNCH(false, 2); break;
        case 3: TG_NF_LAUNCH(false, 3); break;
        case 4: TG_NF_LAUNCH(true, 0); break;
        case 5: TG_NF_LAUNCH(true, 1); break;
        case 6: TG_NF_LAUNCH(true, 2); break;
        default: TG_NF_LAUNCH(true, 3); break;
      }
    } else {
      TG_CHECK(c <= 256, TG_ENOSUP, "tg_norm_act_fwd_partials: scalar path needs c <= 256 (got %d)", c);
      if (z_pooled)
        hipLaunchKernelGGL((norm_act_fwd_part_kernel<T, 1, true>), dim3(chunks2, n), dim3(256), lds, (hipStream_t)stream,
                           (const T*)y, partials, chunks, mean, rstd, gamma, beta, gamma2, beta2, split, (T*)z,
                           (T*)z_pooled, w, pn_scale, hw, c, flags, alpha, in_eps, pn_eps, ppb);
      else
        hipLaunchKernelGGL((norm_act_fwd_part_kernel<T, 1>), dim3(chunks2, n), dim3(256), lds, (hipStream_t)stream,
                           (const T*)y, partials, chunks, mean, rstd, gamma, beta, gamma2, beta2, split, (T*)z,
                           (T*)nullptr, w, pn_scale, hw, c, flags, alpha, in_eps, pn_eps, ppb);
    }
  });
  TG_LAUNCH_CHECK("tg_norm_act_fwd_partials");
  return TG_OK;
}

int tg_norm_act_fwd_partials(const void* y, const float* partials, float* mean, float* rstd, const float* gamma,
                             const float* beta, const float* gamma2, const float* beta2, int split, void* z,
                             void* z_pooled, float* pn_scale, int n, int h, int w, int c, int flags, float alpha,
                             float in_eps, float pn_eps, int dtype, void* stream) {
  return norm_act_fwd_partials_impl(y, partials, 0, mean, rstd, gamma, beta, gamma2, beta2, split, z, z_pooled, pn_scale, n, h,
                                    w, c, flags & ~NF_ZEROSHIFT, alpha, in_eps, pn_eps, dtype, stream);
}

// The same pass fed by the partial sums a conv wrote from its epilogue (tg_conv2d_fwd_stats / tg_conv2d_upcat_fwd_stats:
// [n][part_chunks][2][c], unshifted): no statistics read of y at all.
int tg_norm_act_fwd_conv_stats(const void* y, const float* partials, int part_chunks, float* mean, float* rstd,
                               const float* gamma, const float* beta, const float* gamma2, const float* beta2, int split,
                               void* z, void* z_pooled, float* pn_scale, int n, int h, int w, int c, int flags, float alpha,
                               float in_eps, float pn_eps, int dtype, void* stream) {
  TG_CHECK(part_chunks > 0, TG_EINVAL, "tg_norm_act_fwd_conv_stats: part_chunks must be positive");
  return norm_act_fwd_partials_impl(y, partials, part_chunks, mean, rstd, gamma, beta, gamma2, beta2, split, z, z_pooled,
                                    pn_scale, n, h, w, c, flags | NF_ZEROSHIFT, alpha, in_eps, pn_eps, dtype, stream);
}

int tg_instance_norm_stats(const void* y, float* mean, float* rstd, int n, int h, int w, int c, float eps, int dtype,
                           void* stream) {
  TG_CHECK(y && mean && rstd && n > 0 && h > 0 && w > 0 && c > 0, TG_EINVAL, "tg_instance_norm_stats: bad arguments");
  TG_CHECK(c <= 256 * 8, TG_ENOSUP, "tg_instance_norm_stats: c=%d too large", c);
  hipStream_t s = (hipStream_t)stream;
  const int hw = h * w;
  {
    int rc = tg_zero_async(mean, (size_t)n * c * sizeof(float), rstd, (size_t)n * c * sizeof(float), s);
    if (rc) return rc;
  }
  int chunks = (1024 + n - 1) / n;                      // ~1024 blocks in total
  int ppb = (hw + chunks - 1) / chunks;
  if (ppb < 64) ppb = 64;
  chunks = (hw + ppb - 1) / ppb;
  TG_DISPATCH_DTYPE(dtype, "tg_instance_norm_stats", {
    const int V = pick_v<T>(c);
    TG_CHECK(c / V <= 256, TG_ENOSUP, "tg_instance_norm_stats: c=%d not supported", c);
    const size_t lds = 2 * (size_t)c * sizeof(float);
    if (exact_grid<T>()) {      // one workgroup per image: its sums do not meet another workgroup's in an atomic
      chunks = 1;
      ppb = hw;
    }
    if (V == 1)
      hipLaunchKernelGGL((in_stats_partial<T, 1>), dim3(chunks, n), dim3(256), lds, s, (const T*)y, mean, rstd, hw, c, ppb);
    else
      hipLaunchKernelGGL((in_stats_partial<T, Vec16<T>::N>), dim3(chunks, n), dim3(256), lds, s, (const T*)y, mean, rstd,
                         hw, c, ppb);
    hipLaunchKernelGGL(in_stats_final<T>, dim3((n * c + 255) / 256), dim3(256), 0, s, (const T*)y, mean, rstd, n, hw, c,
                       eps);
  });
  TG_LAUNCH_CHECK("tg_instance_norm_stats");
  return TG_OK;
}

int tg_norm_act_fwd(const void* y, const float* mean, const float* rstd, const float* gamma, const float* beta,
                    const float* gamma2, const float* beta2, int split, int per_image_params, void* z, float* pn_scale,
                    int n, int h, int w, int c, int flags, float alpha, float pn_eps, int dtype, void* stream) {
  const int pstride = per_image_params ? c : 0;
  TG_CHECK(y && mean && rstd && gamma && beta && z && n > 0 && h > 0 && w > 0 && c > 0, TG_EINVAL,
           "tg_norm_act_fwd: bad arguments");
  if (!gamma2 || !beta2) split = n;
  TG_CHECK(split >= 0 && split <= n, TG_EINVAL, "tg_norm_act_fwd: split %d outside [0, %d]", split, n);
  const int64_t npix = (int64_t)n * h * w;
  TG_DISPATCH_DTYPE(dtype, "tg_norm_act_fwd", {
    constexpr int VN = Vec16<T>::N;
    const bool vec = (c % VN == 0) && pow2(c / VN) && c / VN <= 64;
    if (flags & NF_PIXNORM) {
      TG_CHECK(vec, TG_ENOSUP, "tg_norm_act_fwd: pixel norm needs c (%d) = %d * 2^k <= %d", c, VN, 64 * VN);
      TG_CHECK(pn_scale, TG_EINVAL, "tg_norm_act_fwd: pixel norm needs pn_scale");
    }
    if (vec) {
      // grid stride must be a multiple of the group size so pixel groups stay in one wave: 256 % (c/VN) == 0 holds
      hipLaunchKernelGGL((norm_act_fwd_kernel<T, VN>), dim3(tg_grid_for(npix * (c / VN), 256)), dim3(256), 0,
                         (hipStream_t)stream, (const T*)y, mean, rstd, gamma, beta, gamma2, beta2, split, pstride, (T*)z,
                         pn_scale, npix, h * w, c, flags, alpha, pn_eps);
    } else {
      hipLaunchKernelGGL((norm_act_fwd_kernel<T, 1>), dim3(tg_grid_for(npix * c, 256)), dim3(256), 0, (hipStream_t)stream,
                         (const T*)y, mean, rstd, gamma, beta, gamma2, beta2, split, pstride, (T*)z, pn_scale, npix, h * w,
                         c, flags, alpha, pn_eps);
    }
  });
  TG_LAUNCH_CHECK("tg_norm_act_fwd");
  return TG_OK;
}

// the two backward launches of the vector path with the LeakyReLU / pixel-norm bits FL_ fixed at compile time (uses the
// locals of tg_norm_act_bwd)
#define TG_NB_LAUNCH(FL_)                                                                                                  \
  do {                                                                                                                     \
    hipLaunchKernelGGL((norm_act_bwd1_kernel<T, VN, exact_path<T>() ? -1 : FL_>), dim3(chunks, n), dim3(256), lds, s, (const T*)gz, \
                       (const T*)gz_pooled, w, (const T*)y, pn_scale, mean, rstd, gamma, beta, gamma2, beta2, split,       \
                       pstride, sums, hw, c, flags, alpha, ppb);                                                           \
    hipLaunchKernelGGL((norm_act_bwd2_part_kernel<T, VN, exact_path<T>() ? -1 : FL_>), dim3(chunks2, n), dim3(256), lds, s, (const T*)gz, \
                       (const T*)gz_pooled, w, (const T*)y, pn_scale, mean, rstd, gamma, beta, gamma2, beta2, split,       \
                       pstride, (T*)gy, sums, chunks, ggamma, gbeta, ggamma2, gbeta2, sink, hw, c, flags, alpha, ppb2);    \
  } while (0)

int tg_norm_act_bwd(const void* gz, const void* gz_pooled, const void* y, const float* pn_scale, const float* mean,
                    const float* rstd,
                    const float* gamma, const float* beta, const float* gamma2, const float* beta2, int split, void* gy,
                    int per_image_params, float* ggamma, float* gbeta, float* ggamma2, float* gbeta2, float* sums, int n,
                    int h, int w, int c, int flags, float alpha, int accumulate, int dtype, void* stream) {
  const int pstride = per_image_params ? c : 0;
  TG_CHECK((gz || gz_pooled) && y && mean && rstd && gamma && beta && gy && sums && n > 0 && h > 0 && w > 0 && c > 0,
           TG_EINVAL, "tg_norm_act_bwd: bad arguments");
  TG_CHECK(!gz_pooled || (h % 2 == 0 && w % 2 == 0), TG_EINVAL, "tg_norm_act_bwd: pooled gradient needs even h, w");
  if (!gamma2 || !beta2) split = n;
  TG_CHECK(split >= 0 && split <= n, TG_EINVAL, "tg_norm_act_bwd: split %d outside [0, %d]", split, n);
  hipStream_t s = (hipStream_t)stream;
  const int hw = h * w;
  const int64_t npix = (int64_t)n * hw;
  int chunks, ppb;
  norm_chunks(n, hw, &chunks, &ppb);
  int chunks2 = (2048 + n - 1) / n;
  int ppb2 = (hw + chunks2 - 1) / chunks2;
  if (ppb2 < min_ppb(hw)) ppb2 = min_ppb(hw);
  chunks2 = (hw + ppb2 - 1) / ppb2;
  const bool want_params = ggamma || gbeta || ggamma2 || gbeta2;
  TG_CHECK(!pstride || !accumulate, TG_EINVAL, "tg_norm_act_bwd: per-image parameter gradients are written, not added");
  int sink = (want_params && accumulate) ? 1 : 0;            // block (0, n) of pass 2 adds the image's sums
  const size_t lds = 2 * (size_t)c * sizeof(float);
  TG_DISPATCH_DTYPE(dtype, "tg_norm_act_bwd", {
    if (sink && exact_grid<T>()) sink = 2;                   // ... in image order, by the domain's first image
    constexpr int VN = Vec16<T>::N;
    const bool vec = (c % VN == 0) && pow2(c / VN) && c / VN <= 64;
    if (flags & NF_PIXNORM) {
      TG_CHECK(vec && pn_scale, TG_ENOSUP, "tg_norm_act_bwd: pixel norm needs c (%d) = %d * 2^k and pn_scale", c, VN);
    }
    if (vec) {
      switch (flags & 3) {
        case 0: TG_NB_LAUNCH(0); break;
        case 1: TG_NB_LAUNCH(1); break;
        case 2: TG_NB_LAUNCH(2); break;
        default: TG_NB_LAUNCH(3); break;
      }
    } else {
      TG_CHECK(c <= 256, TG_ENOSUP, "tg_norm_act_bwd: scalar path needs c <= 256 (got %d)", c);
      hipLaunchKernelGGL((norm_act_bwd1_kernel<T, 1>), dim3(chunks, n), dim3(256), lds, s, (const T*)gz,
                         (const T*)gz_pooled, w, (const T*)y, pn_scale, mean, rstd, gamma, beta, gamma2, beta2, split, pstride, sums, hw, c, flags, alpha, ppb);
      hipLaunchKernelGGL((norm_act_bwd2_part_kernel<T, 1>), dim3(chunks2, n), dim3(256), lds, s, (const T*)gz,
                         (const T*)gz_pooled, w, (const T*)y, pn_scale, mean, rstd, gamma, beta, gamma2, beta2, split,
                         pstride, (T*)gy, sums, chunks, ggamma, gbeta, ggamma2, gbeta2, sink, hw, c, flags, alpha, ppb2);
    }
  });
  if (want_params && !sink && !pstride)
    hipLaunchKernelGGL(norm_param_grads, dim3((c + 255) / 256, split < n ? 2 : 1), dim3(256), 0, s, sums, chunks, ggamma,
                       gbeta, ggamma2, gbeta2, split, n, c, 0);
  TG_LAUNCH_CHECK("tg_norm_act_bwd");
  return TG_OK;
}

static int lrelu_bwd_launch(const char* who, const void* gz, const void* gzp, int hw, int wdim, const void* z, void* gy,
                            float* gbias, int64_t npix, int c, float alpha, int accumulate, int dtype, hipStream_t s,
                            bool z_is_sign_bits = false) {
  if (gbias && !accumulate) {
    int rc = tg_zero_async(gbias, (size_t)c * sizeof(float), nullptr, 0, s);
    if (rc) return rc;
  }
  TG_DISPATCH_DTYPE(dtype, who, {
    const int V = pick_v<T>(c);
    TG_CHECK(c / V <= 256, TG_ENOSUP, "%s: c=%d not supported", who, c);
    const int lanes = 256 / (c / V);
    // few, fat workgroups when a bias gradient is produced: every workgroup ends with c global atomics (measured: a cap
    // of 512 / 256 workgroups is slower at the 256x256 layers -- 124 / 199 vs 91 us at n48 c32 -- and equal below)
    const int blocks = gbias ? tg_grid_for(npix, lanes * 16, 1024) : tg_grid_for(npix, lanes * 4, 2048);
    const size_t lds = (size_t)c * sizeof(float);
    // exact path: the element-wise part on the full grid, the bias gradient by one workgroup over what it wrote
    float* gb_here = (exact_grid<T>() && blocks > 1) ? nullptr : gbias;
    if (z_is_sign_bits) {
      if constexpr (sizeof(T) == 2) {
        TG_CHECK(V == 8, TG_ENOSUP, "%s: sign bits need c %% 8 == 0 (c = %d)", who, c);
        hipLaunchKernelGGL((lrelu_bwd_bias_kernel<T, 8, true>), dim3(blocks), dim3(256), lds, s, (const T*)gz, (const T*)gzp,
                           hw, wdim, (const T*)z, (T*)gy, gb_here, npix, c, alpha);
      } else {
        TG_CHECK(false, TG_ENOSUP, "%s: sign bits are a feature of the 16-bit storage types", who);
      }
    } else if (V == 1)
      hipLaunchKernelGGL((lrelu_bwd_bias_kernel<T, 1>), dim3(blocks), dim3(256), lds, s, (const T*)gz, (const T*)gzp, hw,
                         wdim, (const T*)z, (T*)gy, gb_here, npix, c, alpha);
    else
      hipLaunchKernelGGL((lrelu_bwd_bias_kernel<T, Vec16<T>::N>), dim3(blocks), dim3(256), lds, s, (const T*)gz,
                         (const T*)gzp, hw, wdim, (const T*)z, (T*)gy, gb_here, npix, c, alpha);
    if (gbias && !gb_here) {
      if (V == 1)
        hipLaunchKernelGGL((channel_sum_kernel<T, 1>), dim3(1), dim3(256), lds, s, (const T*)gy, gbias, npix, c);
      else
        hipLaunchKernelGGL((channel_sum_kernel<T, Vec16<T>::N>), dim3(1), dim3(256), lds, s, (const T*)gy, gbias, npix, c);
    }
  });
  hipError_t e__ = hipGetLastError();
  if (e__ != hipSuccess) {
    tg_set_error("%s: launch failed: %s", who, hipGetErrorString(e__));
    return TG_ELAUNCH;
  }
  return TG_OK;
}

int tg_lrelu_bwd_bias(const void* gz, const void* z, void* gy, float* gbias, int64_t npix, int c, float alpha,
                      int accumulate, int dtype, void* stream) {
  TG_CHECK(gz && z && gy && gbias && npix > 0 && c > 0, TG_EINVAL, "tg_lrelu_bwd_bias: bad arguments");
  return lrelu_bwd_launch("tg_lrelu_bwd_bias", gz, nullptr, 1, 1, z, gy, gbias, npix, c, alpha, accumulate, dtype,
                          (hipStream_t)stream);
}

int tg_lrelu_pool_bwd(const void* gz, const void* gz_pooled, const void* z, void* gy, float* gbias, int n, int h, int w,
                      int c, float alpha, int accumulate, int dtype, void* stream) {
  TG_CHECK((gz || gz_pooled) && z && gy && n > 0 && h > 0 && w > 0 && c > 0, TG_EINVAL, "tg_lrelu_pool_bwd: bad arguments");
  TG_CHECK(!gz_pooled || (h % 2 == 0 && w % 2 == 0), TG_EINVAL, "tg_lrelu_pool_bwd: pooled gradient needs even h, w");
  if (!gz_pooled)
    return lrelu_bwd_launch("tg_lrelu_pool_bwd", gz, nullptr, 1, 1, z, gy, gbias, (int64_t)n * h * w, c, alpha, accumulate,
                            dtype, (hipStream_t)stream);
  return lrelu_bwd_launch("tg_lrelu_pool_bwd", gz, gz_pooled, h * w, w, z, gy, gbias, (int64_t)n * h * w, c, alpha,
                          accumulate, dtype, (hipStream_t)stream);
}

int tg_lrelu_pool_bwd_signs(const void* gz_pooled, const void* z_signs, void* gy, float* gbias, int n, int h, int w, int c,
                            float alpha, int accumulate, int dtype, void* stream) {
  TG_CHECK(gz_pooled && z_signs && gy && n > 0 && h > 0 && w > 0 && c > 0, TG_EINVAL, "tg_lrelu_pool_bwd_signs: bad arguments");
  TG_CHECK(h % 2 == 0 && w % 2 == 0 && c % 8 == 0, TG_EINVAL, "tg_lrelu_pool_bwd_signs: needs even h, w and c %% 8 == 0");
  TG_CHECK(dtype == TG_BF16 || dtype == TG_F16, TG_ENOSUP, "tg_lrelu_pool_bwd_signs: 16-bit storage only");
  return lrelu_bwd_launch("tg_lrelu_pool_bwd_signs", nullptr, gz_pooled, h * w, w, z_signs, gy, gbias, (int64_t)n * h * w, c,
                          alpha, accumulate, dtype, (hipStream_t)stream, true);
}

int tg_channel_sum_ordered(const void* g, float* out, int64_t npix, int c, int accumulate, float* ws, size_t ws_floats,
                           int dtype, void* stream) {
  TG_CHECK(g && out && ws && npix > 0 && c > 0, TG_EINVAL, "tg_channel_sum_ordered: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  int blocks = 0;
  TG_DISPATCH_DTYPE(dtype, "tg_channel_sum_ordered", {
    const int V = pick_v<T>(c);
    TG_CHECK(c / V <= 256, TG_ENOSUP, "tg_channel_sum_ordered: c=%d not supported", c);
    const int lanes = 256 / (c / V);
    blocks = tg_grid_for(npix, lanes * 8, 512);
    if ((size_t)blocks * c > ws_floats) blocks = (int)(ws_floats / c);
    TG_CHECK(blocks >= 1, TG_EINVAL, "tg_channel_sum_ordered: workspace of %zu floats cannot hold one row of %d", ws_floats, c);
    const size_t lds = (size_t)c * sizeof(float);
    if (V == 1)
      hipLaunchKernelGGL((channel_sum_kernel<T, 1, true>), dim3(blocks), dim3(256), lds, s, (const T*)g, ws, npix, c);
    else
      hipLaunchKernelGGL((channel_sum_kernel<T, Vec16<T>::N, true>), dim3(blocks), dim3(256), lds, s, (const T*)g, ws, npix, c);
  });
  hipLaunchKernelGGL(ordered_rows_sum_kernel, dim3((c + 255) / 256), dim3(256), 0, s, ws, blocks, c, out, accumulate, 1.f);
  TG_LAUNCH_CHECK("tg_channel_sum_ordered");
  return TG_OK;
}

int tg_channel_sum(const void* g, float* out, int64_t npix, int c, int accumulate, int dtype, void* stream) {
  TG_CHECK(g && out && npix > 0 && c > 0, TG_EINVAL, "tg_channel_sum: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate) {
    int rc = tg_zero_async(out, (size_t)c * sizeof(float), nullptr, 0, s);
    if (rc) return rc;
  }
  TG_DISPATCH_DTYPE(dtype, "tg_channel_sum", {
    const int V = pick_v<T>(c);
    TG_CHECK(c / V <= 256, TG_ENOSUP, "tg_channel_sum: c=%d not supported", c);
    const int lanes = 256 / (c / V);
    const int blocks = exact_grid<T>() ? 1 : tg_grid_for(npix, lanes * 8, 1024);
    const size_t lds = (size_t)c * sizeof(float);
    if (V == 1)
      hipLaunchKernelGGL((channel_sum_kernel<T, 1>), dim3(blocks), dim3(256), lds, s, (const T*)g, out, npix, c);
    else
      hipLaunchKernelGGL((channel_sum_kernel<T, Vec16<T>::N>), dim3(blocks), dim3(256), lds, s, (const T*)g, out, npix, c);
  });
  TG_LAUNCH_CHECK("tg_channel_sum");
  return TG_OK;
}

}  // extern "C"
