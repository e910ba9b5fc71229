// Bandwidth-bound pointwise / resampling kernels of the TwinGAN hot path (NHWC, fp32 or bf16
// storage, fp32 math, 16-byte vector accesses wherever the channel count allows).
#include "tg_common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// generic vectorised elementwise driver: f(i, vec...) over numel elements
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void axpby_kernel(const T* __restrict__ x, const T* __restrict__ y, T* __restrict__ out, int64_t numel, float a,
                             float b) {
  constexpr int V = Vec16<T>::N;
  const int64_t nvec = numel / V;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    Vec16<T> vx = ldv(x + i * V), vo;
    if (y) {
      Vec16<T> vy = ldv(y + i * V);
#pragma unroll
      for (int j = 0; j < V; ++j) vo.set(j, a * vx.get(j) + b * vy.get(j));
    } else {
#pragma unroll
      for (int j = 0; j < V; ++j) vo.set(j, a * vx.get(j));
    }
    stv(out + i * V, vo);
  }
  for (int64_t i = nvec * V + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += stride)
    st(out + i, a * ld(x + i) + (y ? b * ld(y + i) : 0.f));
}

// ---- row blocks of a batch put together: dst[dst_off + i] = sum_k src[k][i] (fp32 sum, rounded once), zeros without a source.
// One launch for every block of a destination: the copies of a concatenation along N, the repeat of a batch, and -- the
// backward of tensors read through several row ranges -- the sum of the gradients that cover each block.
struct RowsJobs {
  TgRowsJob j[TG_ROWS_MAX_JOBS];
};
template <typename T, bool VEC>
__global__ void rows_assemble_kernel(RowsJobs jobs, T* __restrict__ dst) {
  const TgRowsJob& jb = jobs.j[blockIdx.y];
  const T* s0 = (const T*)jb.src[0];
  const T* s1 = (const T*)jb.src[1];
  const T* s2 = (const T*)jb.src[2];
  const T* s3 = (const T*)jb.src[3];
  T* d = dst + jb.dst_off;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if constexpr (VEC) {
    constexpr int V = Vec16<T>::N;
    const int64_t nvec = jb.numel / V;
    for (int64_t i = t0; i < nvec; i += stride) {
      Vec16<T> vo;
      if (!s0) {
#pragma unroll
        for (int j = 0; j < V; ++j) vo.set(j, 0.f);
      } else if (!s1) {
        vo = ldv(s0 + i * V);
      } else {
        float acc[V];
        Vec16<T> a = ldv(s0 + i * V), b = ldv(s1 + i * V);
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] = a.get(j) + b.get(j);
        if (s2) {
          a = ldv(s2 + i * V);
#pragma unroll
          for (int j = 0; j < V; ++j) acc[j] += a.get(j);
        }
        if (s3) {
          a = ldv(s3 + i * V);
#pragma unroll
          for (int j = 0; j < V; ++j) acc[j] += a.get(j);
        }
#pragma unroll
        for (int j = 0; j < V; ++j) vo.set(j, acc[j]);
      }
      stv(d + i * V, vo);
    }
  } else {
    for (int64_t i = t0; i < jb.numel; i += stride) {
      float acc = 0.f;
      if (s0) acc = ld(s0 + i);
      if (s1) acc += ld(s1 + i);
      if (s2) acc += ld(s2 + i);
      if (s3) acc += ld(s3 + i);
      st(d + i, acc);
    }
  }
}

// ---- U[0, 1) draws: Philox4x32-10 keyed by (seed, draw counter), one 4-word block per four outputs.  The draw counter is
// a DEVICE word the kernel itself advances (the last workgroup to finish), so a captured launch draws new numbers on every
// replay; state[1] is that rendezvous' ticket and is left at 0.
__device__ __forceinline__ void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1) {
  const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
  const unsigned h0 = (unsigned)(p0 >> 32), l0 = (unsigned)p0, h1 = (unsigned)(p1 >> 32), l1 = (unsigned)p1;
  c[0] = h1 ^ c[1] ^ k0;
  c[1] = l1;
  c[2] = h0 ^ c[3] ^ k1;
  c[3] = l0;
}
__global__ void uniform_kernel(float* __restrict__ out, int64_t n, unsigned seed_lo, unsigned seed_hi,
                               unsigned* __restrict__ state, float lo, float scale) {
  const unsigned draw = __hip_atomic_load(state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q * 4 < n; q += stride) {
    unsigned c[4] = {(unsigned)q, (unsigned)(q >> 32), draw, 0u};
    unsigned k0 = seed_lo, k1 = seed_hi;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      philox_round(c, k0, k1);
      k0 += 0x9E3779B9u;
      k1 += 0xBB67AE85u;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (q * 4 + j < n) out[q * 4 + j] = lo + scale * ((float)(c[j] >> 8) * (1.f / 16777216.f));      // 24 bits: [0, 1)
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    const unsigned ticket = atomicAdd(state + 1, 1u);
    if (ticket == gridDim.x - 1) {      // every workgroup has read the counter: advance it, reset the ticket
      __hip_atomic_store(state + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(state, draw + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

template <typename T>
__global__ void lrelu_bwd_kernel(const T* __restrict__ gz, const T* __restrict__ z, T* __restrict__ gy, int64_t numel,
                                 float alpha) {
  constexpr int V = Vec16<T>::N;
  const int64_t nvec = numel / V;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    Vec16<T> vg = ldv(gz + i * V), vz = ldv(z + i * V), vo;
#pragma unroll
    for (int j = 0; j < V; ++j) vo.set(j, vg.get(j) * (vz.get(j) > 0.f ? 1.f : alpha));
    stv(gy + i * V, vo);
  }
  for (int64_t i = nvec * V + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += stride)
    st(gy + i, ld(gz + i) * (ld(z + i) > 0.f ? 1.f : alpha));
}

template <typename T>
__global__ void bias_lrelu_kernel(const T* __restrict__ y, const float* __restrict__ bias, T* __restrict__ z,
                                  int64_t numel, int c, float alpha) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  constexpr int V = Vec16<T>::N;
  if (c % V == 0) {
    const int64_t nvec = numel / V;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
      const int ch = (int)((i * V) % c);
      Vec16<T> vy = ldv(y + i * V), vo;
#pragma unroll
      for (int j = 0; j < V; ++j) vo.set(j, lrelu_f(vy.get(j) + (bias ? bias[ch + j] : 0.f), alpha));
      stv(z + i * V, vo);
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += stride)
      st(z + i, lrelu_f(ld(y + i) + (bias ? bias[i % c] : 0.f), alpha));
  }
}

template <typename T>
__global__ void fill_scaled_kernel(T* __restrict__ out, const float* __restrict__ scalar, float value, int64_t numel) {
  const float v = value * (scalar ? scalar[0] : 1.f);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (int64_t)gridDim.x * blockDim.x)
    st(out + i, v);
}

template <typename TS, typename TD>
__global__ void cast_kernel(const TS* __restrict__ s, TD* __restrict__ d, int64_t numel) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += stride) st(d + i, ld(s + i));
}

// out[b, :] = x + alpha[b] * (y - x)
template <typename T>
__global__ void sample_lerp_kernel(const T* __restrict__ x, const T* __restrict__ y, const float* __restrict__ alpha,
                                   T* __restrict__ out, int64_t per, int64_t numel) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += stride) {
    const float a = alpha[i / per], vx = ld(x + i);
    st(out + i, vx + a * (ld(y + i) - vx));
  }
}

template <typename T>
__global__ void sample_scale_kernel(const T* __restrict__ x, const float* __restrict__ coef,
                                    const float* __restrict__ scalar, T* __restrict__ out, int64_t per, int64_t numel) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const float sc = scalar ? scalar[0] : 1.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += stride)
    st(out + i, ld(x + i) * coef[i / per] * sc);
}

// gdrop, mode 'prop' (libs/gdrop.py:20-36): out[n, p, c] = x[n, p, c] * (noise[n, c] * coef + 1) with
// coef = strength * sqrt(c_logical); the strength is a DEVICE scalar (the `gdrop_strength` variable of the controller,
// image_generation.py:563-585) or a constant.  c may be channel-padded (the minibatch-stddev tensor): noise has c entries
// per image, the pad channels of x are zero.  One thread per element: this is an off-by-default layer of the plain PGGAN
// discriminator, a single pass at the tensor's byte rate.
template <typename T>
__global__ void gdrop_kernel(const T* __restrict__ x, const float* __restrict__ noise, const float* __restrict__ strength_dev,
                             float strength, float sqrt_c, T* __restrict__ out, int64_t per_image, int c, int64_t numel) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const float coef = (strength_dev ? strength_dev[0] : strength) * sqrt_c;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += stride) {
    const int64_t img = i / per_image;
    const int ch = (int)(i % c);
    st(out + i, ld(x + i) * fmaf(noise[img * c + ch], coef, 1.f));
  }
}

// ------------------------------------------------------------------------------------------------
// 2x nearest upsample (+ channel concat), its backward; 2x2 pool and its backward.
// One thread per 16-byte channel vector of one OUTPUT pixel (scalar path when c % V != 0).
// ------------------------------------------------------------------------------------------------
template <typename T, int V>
__global__ void upsample_concat_fwd_kernel(const T* __restrict__ x0, const T* __restrict__ x1, T* __restrict__ out, int n,
                                           int h, int w, int c0, int c1, int gsz, unsigned perm) {
  const int c = c0 + c1, cv = c / V, c0v = c0 / V;
  const int64_t total = (int64_t)n * (2 * h) * (2 * w) * cv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int v = (int)(i % cv);
    int64_t p = i / cv;
    const int ox = (int)(p % (2 * w));
    p /= 2 * w;
    const int oy = (int)(p % (2 * h));
    const int in_ = (int)(p / (2 * h));
    // skip source image: identity, or group permutation (out group k of gsz images reads x1 group perm[k])
    const int i1 = gsz ? (int)((perm >> (8 * (in_ / gsz))) & 0xffu) * gsz + in_ % gsz : in_;
    const T* src = (v < c0v) ? x0 + (((int64_t)in_ * h + (oy >> 1)) * w + (ox >> 1)) * c0 + v * V
                             : x1 + (((int64_t)i1 * 2 * h + oy) * (2 * w) + ox) * c1 + (v - c0v) * V;
    T* dst = out + (((int64_t)in_ * 2 * h + oy) * (2 * w) + ox) * c + v * V;
    if (V == 1)
      *dst = *src;
    else
      stv(dst, ldv(src));
  }
}

template <typename T, int V>
__global__ void upsample_concat_bwd_kernel(const T* __restrict__ go, T* __restrict__ g0, T* __restrict__ g1, int n, int h,
                                           int w, int c0, int c1, int gsz, unsigned perm, int n1) {
  // g0: one thread per (input pixel, vector) sums its 2x2 block; g1: copy of the tail channels (summed over
  // the output groups that read the same skip image when a group permutation is in use).
  const int c = c0 + c1, c0v = c0 / V, c1v = c1 / V;
  const int64_t t0 = g0 ? (int64_t)n * h * w * c0v : 0;
  const int64_t t1 = g1 ? (int64_t)n1 * 4 * h * w * c1v : 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < t0 + t1; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < t0) {
      const int v = (int)(i % c0v);
      int64_t p = i / c0v;
      const int x = (int)(p % w);
      p /= w;
      const int y = (int)(p % h);
      const int in_ = (int)(p / h);
      float acc[V];
#pragma unroll
      for (int j = 0; j < V; ++j) acc[j] = 0.f;
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          const T* src = go + (((int64_t)in_ * 2 * h + 2 * y + dy) * (2 * w) + 2 * x + dx) * c + v * V;
          if (V == 1) {
            acc[0] += ld(src);
          } else {
            Vec16<T> s = ldv(src);
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] += s.get(j);
          }
        }
      T* dst = g0 + (((int64_t)in_ * h + y) * w + x) * c0 + v * V;
      if (V == 1) {
        st(dst, acc[0]);
      } else {
        Vec16<T> o;
#pragma unroll
        for (int j = 0; j < V; ++j) o.set(j, acc[j]);
        stv(dst, o);
      }
    } else {
      const int64_t k = i - t0;
      const int v = (int)(k % c1v);
      const int64_t p = k / c1v;      // skip-tensor pixel linear index (image-major)
      T* dst = g1 + p * c1 + v * V;
      if (!gsz) {
        const T* src = go + p * c + c0 + v * V;
        if (V == 1)
          *dst = *src;
        else
          stv(dst, ldv(src));
      } else {
        const int64_t ipx = (int64_t)4 * h * w;
        const int j = (int)(p / ipx);                 // skip image
        const int64_t q = p - (int64_t)j * ipx;
        const int sg = j / gsz, jj = j - sg * gsz;
        float acc[V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.f;
        for (int og = 0; og < n / gsz; ++og) {
          if ((int)((perm >> (8 * og)) & 0xffu) != sg) continue;
          const T* src = go + (((int64_t)(og * gsz + jj)) * ipx + q) * c + c0 + v * V;
          if (V == 1) {
            acc[0] += ld(src);
          } else {
            Vec16<T> sv = ldv(src);
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] += sv.get(e);
          }
        }
        if (V == 1) {
          st(dst, acc[0]);
        } else {
          Vec16<T> o;
#pragma unroll
          for (int e = 0; e < V; ++e) o.set(e, acc[e]);
          stv(dst, o);
        }
      }
    }
  }
}

template <typename T, int V>
__global__ void pool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int n, int h, int w, int c, float scale) {
  const int cv = c / V, ho = h / 2, wo = w / 2;
  const int64_t total = (int64_t)n * ho * wo * cv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int v = (int)(i % cv);
    int64_t p = i / cv;
    const int ox = (int)(p % wo);
    p /= wo;
    const int oy = (int)(p % ho);
    const int in_ = (int)(p / ho);
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const T* src = x + (((int64_t)in_ * h + 2 * oy + dy) * w + 2 * ox + dx) * c + v * V;
        if (V == 1) {
          acc[0] += ld(src);
        } else {
          Vec16<T> s = ldv(src);
#pragma unroll
          for (int j = 0; j < V; ++j) acc[j] += s.get(j);
        }
      }
    T* dst = y + (((int64_t)in_ * ho + oy) * wo + ox) * c + v * V;
    if (V == 1) {
      st(dst, acc[0] * scale);
    } else {
      Vec16<T> o;
#pragma unroll
      for (int j = 0; j < V; ++j) o.set(j, acc[j] * scale);
      stv(dst, o);
    }
  }
}

template <typename T, int V>
__global__ void pool_bwd_kernel(const T* __restrict__ gy, T* __restrict__ gx, int n, int h, int w, int c, float scale) {
  const int cv = c / V, ho = h / 2, wo = w / 2;
  const int64_t total = (int64_t)n * h * w * cv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int v = (int)(i % cv);
    int64_t p = i / cv;
    const int x = (int)(p % w);
    p /= w;
    const int y = (int)(p % h);
    const int in_ = (int)(p / h);
    T* dst = gx + (((int64_t)in_ * h + y) * w + x) * c + v * V;
    if ((y >> 1) >= ho || (x >> 1) >= wo) {   // odd trailing row/col (VALID pooling drops it)
      if (V == 1) {
        st(dst, 0.f);
      } else {
        Vec16<T> o;
#pragma unroll
        for (int j = 0; j < V; ++j) o.set(j, 0.f);
        stv(dst, o);
      }
      continue;
    }
    const T* src = gy + (((int64_t)in_ * ho + (y >> 1)) * wo + (x >> 1)) * c + v * V;
    if (V == 1) {
      st(dst, ld(src) * scale);
    } else {
      Vec16<T> s = ldv(src), o;
#pragma unroll
      for (int j = 0; j < V; ++j) o.set(j, s.get(j) * scale);
      stv(dst, o);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// 1x1 convs with a tiny (<= 4) channel count on one side.
// ------------------------------------------------------------------------------------------------
// small-in: y[p, co] = sum_{ci<cin} x[p,ci] * w[ci,co]; thread = (pixel, V-vector of co)
template <typename T, int V>
__global__ void pw_small_in_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                   T* __restrict__ y, int64_t npix, int cin, int cout, int wt, int epi, float alpha,
                                   const T* __restrict__ mask) {
  // mask (optional, y's shape): the ROUNDED output times (mask > 0 ? 1 : alpha) -- the LeakyReluGrad the gradient penalty's
  // second backward applies to what fromRGB's transposed conv hands the first block (tg_pointwise_conv_fwd_masked)
  extern __shared__ float sw[];   // [cin][cout] (+ bias[cout])
  for (int i = threadIdx.x; i < cin * cout; i += blockDim.x) {
    const int ci = i / cout, co = i - ci * cout;
    sw[i] = rnd<T>(wt ? w[co * cin + ci] : w[i]);
  }
  for (int i = threadIdx.x; i < cout; i += blockDim.x) sw[cin * cout + i] = (epi & TG_EPI_BIAS) ? bias[i] : 0.f;
  __syncthreads();
  const int cv = cout / V;
  const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
  if (cv <= 256 && (cv & (cv - 1)) == 0) {
    // fast path (cv divides the thread count): a thread keeps ONE channel vector for its whole life, so its
    // cin x V weights and biases sit in registers and the pixel loop has no division and no LDS traffic
    const int v = (int)(gtid % cv);
    float wr[4][V], br[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
      br[j] = sw[cin * cout + v * V + j];
#pragma unroll
      for (int ci = 0; ci < 4; ++ci) wr[ci][j] = ci < cin ? sw[ci * cout + v * V + j] : 0.f;
    }
    // U pixels in flight per thread (a grid stride apart, so every store instruction of a wave stays one contiguous run):
    // the loads of a pixel are three 2-byte scalars -- with one pixel per trip a CU had a few KB of reads outstanding and
    // the layer ran at 2.5 TB/s of its (write-dominated) bytes.  Loads clamped, not predicated: no branches between them.
    constexpr int U = 4;
    const int64_t pstride = nthreads / cv;
    const int cl = cin - 1;
    for (int64_t p0 = gtid / cv; p0 < npix; p0 += pstride * U) {
      float xin[U][4];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        int64_t p = p0 + u * pstride;
        p = p < npix ? p : npix - 1;
#pragma unroll
        for (int ci = 0; ci < 4; ++ci) xin[u][ci] = ld(x + p * cin + (ci < cl ? ci : cl));      // ci >= cin: weight 0
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t p = p0 + u * pstride;
        if (p >= npix) break;
        float acc[V];
#pragma unroll
        for (int j = 0; j < V; ++j) {
          float a = br[j];
#pragma unroll
          for (int ci = 0; ci < 4; ++ci) a = fmaf(xin[u][ci], wr[ci][j], a);
          if (epi & TG_EPI_LRELU) a = lrelu_f(a, alpha);
          acc[j] = a;
        }
        T* dst = y + p * cout + v * V;
        if (V == 1) {
          st(dst, mask ? rnd<T>(acc[0]) * (ld(mask + p * cout + v) > 0.f ? 1.f : alpha) : acc[0]);
        } else {
          Vec16<T> o;
#pragma unroll
          for (int j = 0; j < V; ++j) o.set(j, acc[j]);
          if (mask) {
            const Vec16<T> m = ldv(mask + p * cout + v * V);
#pragma unroll
            for (int j = 0; j < V; ++j) o.set(j, o.get(j) * (m.get(j) > 0.f ? 1.f : alpha));
          }
          stv(dst, o);
        }
      }
    }
    return;
  }
  const int64_t total = npix * cv;
  for (int64_t i = gtid; i < total; i += nthreads) {
    const int v = (int)(i % cv);
    const int64_t p = i / cv;
    float xin[4];
    for (int ci = 0; ci < cin; ++ci) xin[ci] = ld(x + p * cin + ci);
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float a = 0.f;
      for (int ci = 0; ci < cin; ++ci) a = fmaf(xin[ci], sw[ci * cout + v * V + j], a);
      a += sw[cin * cout + v * V + j];
      if (epi & TG_EPI_LRELU) a = lrelu_f(a, alpha);
      acc[j] = a;
    }
    T* dst = y + p * cout + v * V;
    if (V == 1) {
      st(dst, mask ? rnd<T>(acc[0]) * (ld(mask + p * cout + v) > 0.f ? 1.f : alpha) : acc[0]);
    } else {
      Vec16<T> o;
#pragma unroll
      for (int j = 0; j < V; ++j) o.set(j, acc[j]);
      if (mask) {
        const Vec16<T> m = ldv(mask + p * cout + v * V);
#pragma unroll
        for (int j = 0; j < V; ++j) o.set(j, o.get(j) * (m.get(j) > 0.f ? 1.f : alpha));
      }
      stv(dst, o);
    }
  }
}

// small-out: y[p, co<cout<=4] = sum_ci x[p,ci] * w[ci,co]; thread = pixel, loops over V-vectors of ci
template <typename T, int V>
__global__ void pw_small_out_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                    T* __restrict__ y, int64_t npix, int cin, int cout, int wt, int epi, float alpha) {
  extern __shared__ float sw[];   // [cout][cin]  (transposed for contiguous reads) + bias
  for (int i = threadIdx.x; i < cin * cout; i += blockDim.x) {
    const int co = i / cin, ci = i - co * cin;
    sw[i] = rnd<T>(wt ? w[co * cin + ci] : w[ci * cout + co]);
  }
  for (int i = threadIdx.x; i < cout; i += blockDim.x) sw[cin * cout + i] = (epi & TG_EPI_BIAS) ? bias[i] : 0.f;
  __syncthreads();
  const int civ = cin / V;
  if (V > 1 && civ == 2) {
    // toRGB at full resolution (16 -> 3 channels, 1-4 M pixels): U pixels in flight per thread, a grid stride apart
    constexpr int U = 4;
    const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p0 < npix; p0 += nthreads * U) {
      Vec16<T> xv[U][2];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        int64_t p = p0 + u * nthreads;
        p = p < npix ? p : npix - 1;
        xv[u][0] = ldv(x + p * cin);
        xv[u][1] = ldv(x + p * cin + V);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t p = p0 + u * nthreads;
        if (p >= npix) break;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int v = 0; v < 2; ++v)
          for (int co = 0; co < cout; ++co)
#pragma unroll
            for (int j = 0; j < V; ++j) acc[co] = fmaf(xv[u][v].get(j), sw[co * cin + v * V + j], acc[co]);
        for (int co = 0; co < cout; ++co) {
          float a = acc[co] + sw[cin * cout + co];
          if (epi & TG_EPI_LRELU) a = lrelu_f(a, alpha);
          st(y + p * cout + co, a);
        }
      }
    }
    return;
  }
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (int64_t)gridDim.x * blockDim.x) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int v = 0; v < civ; ++v) {
      float xv[V];
      if (V == 1) {
        xv[0] = ld(x + p * cin + v);
      } else {
        Vec16<T> s = ldv(x + p * cin + v * V);
#pragma unroll
        for (int j = 0; j < V; ++j) xv[j] = s.get(j);
      }
      for (int co = 0; co < cout; ++co)
#pragma unroll
        for (int j = 0; j < V; ++j) acc[co] = fmaf(xv[j], sw[co * cin + v * V + j], acc[co]);
    }
    for (int co = 0; co < cout; ++co) {
      float a = acc[co] + sw[cin * cout + co];
      if (epi & TG_EPI_LRELU) a = lrelu_f(a, alpha);
      st(y + p * cout + co, a);
    }
  }
}

// out[s*os + c*oc] += sum_p small[p, s] * big[p, c];   thread = (pixel lane, V-vector of big channels)
// PART: out[blockIdx.x][s][c] = this workgroup's sums (dense [ns][cb] rows; pw_wgrad_final adds them in workgroup order)
template <typename T, int V, bool PART = false>
__global__ void pw_wgrad_kernel(const T* __restrict__ small, const T* __restrict__ big, float* __restrict__ out,
                                int64_t npix, int ns, int cb, int os, int oc) {
  extern __shared__ float sacc[];   // [ns][cb], then one [ns][cb] slot per wave (pow2 channel-vector counts)
  for (int i = threadIdx.x; i < ns * cb; i += blockDim.x) sacc[i] = 0.f;
  __syncthreads();
  const int cv = cb / V;
  const int lanes = blockDim.x / cv;           // pixel lanes per block
  const int v = threadIdx.x % cv, pl = threadIdx.x / cv;
  float acc[4][V];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int j = 0; j < V; ++j) acc[s][j] = 0.f;
  if (pl < lanes) {
    constexpr int U = 4;      // pixels in flight per thread
    const int64_t stride = (int64_t)gridDim.x * lanes;
    for (int64_t pb = (int64_t)blockIdx.x * lanes + pl; pb < npix; pb += stride * U) {
      float bv[U][V], sv[U][4];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        int64_t p = pb + u * stride;
        p = p < npix ? p : npix - 1;      // clamped, not predicated: no branches between the loads
        if (V == 1) {
          bv[u][0] = ld(big + p * cb + v);
        } else {
          Vec16<T> t = ldv(big + p * cb + v * V);
#pragma unroll
          for (int j = 0; j < V; ++j) bv[u][j] = t.get(j);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) sv[u][s] = s < ns ? ld(small + p * ns + s) : 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (pb + u * stride < npix) {
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < V; ++j) acc[s][j] = fmaf(sv[u][s], bv[u][j], acc[s][j]);
        }
      }
    }
  }
  if (cv <= 64 && (cv & (cv - 1)) == 0) {
    // sum the pixel lanes of a wave that own the same channel vector first: cv lanes per wave touch LDS
    for (int o = cv; o < 64; o <<= 1) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < V; ++j) acc[s][j] += __shfl_xor(acc[s][j], o, 64);
    }
    // the waves' sums go to per-wave slots and are added in wave order: no LDS atomics, a fixed summation order
    float* slot = sacc + (1 + (threadIdx.x >> 6)) * ns * cb;
    if ((int)(threadIdx.x & 63) < cv) {
      for (int s = 0; s < ns; ++s)
#pragma unroll
        for (int j = 0; j < V; ++j) slot[s * cb + v * V + j] = acc[s][j];
    }
    __syncthreads();
    const int nw = blockDim.x >> 6;
    for (int i = threadIdx.x; i < ns * cb; i += blockDim.x) {
      float t = 0.f;
      for (int wv = 0; wv < nw; ++wv) t += sacc[(1 + wv) * ns * cb + i];
      sacc[i] = t;
    }
  } else if (pl < lanes) {
    for (int s = 0; s < ns; ++s)
#pragma unroll
      for (int j = 0; j < V; ++j) atomicAdd(&sacc[s * cb + v * V + j], acc[s][j]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < ns * cb; i += blockDim.x) {
    const int s = i / cb, c = i - s * cb;
    if constexpr (PART) out[(size_t)blockIdx.x * ns * cb + i] = sacc[i];
    else atomicAdd(out + (int64_t)s * os + (int64_t)c * oc, sacc[i]);
  }
}

// The fromRGB filter gradient at full resolution (nets/pggan.py:233-240,395-399: 3 -> 16 channels at 256 x 256, 1-4 M pixels
// per launch).  pw_wgrad_kernel fetches a pixel's three RGB values as three 2-byte loads in EVERY channel-vector thread of
// the pixel -- four memory instructions per 22 bytes: 1.06 TB/s (profiles/r03_z_shapes_eager_step.json).  Here a thread
// owns one 16-byte channel vector of FOUR consecutive pixels: their 12 RGB values are three aligned 8-byte loads, the big
// side four 16-byte loads; U quads in flight.  BIAS: the big side is the layer's output gradient and its per-channel
// pixel sum (the BiasAddGrad of the same layer, a separate tg_channel_sum pass over the same tensor before) rides along
// as a fourth accumulator row.  16-bit storage types, 3 small channels, cb = 8 * 2^k <= 512 channels, npix % 4 == 0.
template <typename T> struct Quad16;
template <> struct Quad16<bf16> { typedef bf16x4 type; };
template <> struct Quad16<f16> { typedef __attribute__((ext_vector_type(4))) _Float16 type; };
template <typename T, bool BIAS>
__global__ __launch_bounds__(1024) void pw_wgrad_rgb4_kernel(const T* __restrict__ small, const T* __restrict__ big,
                                                             float* __restrict__ out, float* __restrict__ gbias,
                                                             int64_t nquad, int cb, int os, int oc) {
  constexpr int V = 8, NS = 3, R = NS + (BIAS ? 1 : 0), U = 2;
  typedef typename Quad16<T>::type q16;
  extern __shared__ float sacc[];   // result [R][cb], then one [R][cb] slot per wave
  const int cv = cb / V;
  const int lanes = blockDim.x / cv;           // quad lanes per block (cv is a power of two <= 64: lanes * cv = blockDim)
  const int v = threadIdx.x % cv, pl = threadIdx.x / cv;
  float acc[R][V];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int j = 0; j < V; ++j) acc[r][j] = 0.f;
  const int64_t stride = (int64_t)gridDim.x * lanes;
  for (int64_t qb = (int64_t)blockIdx.x * lanes + pl; qb < nquad; qb += stride * U) {
    Vec16<T> bv[U][4];
    q16 sw[U][NS];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int64_t q = qb + u * stride;
      q = q < nquad ? q : nquad - 1;      // clamped, not predicated: no branches between the loads
#pragma unroll
      for (int k = 0; k < 4; ++k) bv[u][k] = ldv(big + (4 * q + k) * cb + v * V);
      const q16* sp = reinterpret_cast<const q16*>(small + 4 * q * NS);      // 24 bytes per quad: 8-byte aligned
#pragma unroll
      for (int w = 0; w < NS; ++w) sw[u][w] = sp[w];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (qb + u * stride < nquad) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float sv[NS];
#pragma unroll
          for (int c3 = 0; c3 < NS; ++c3) sv[c3] = (float)sw[u][(k * NS + c3) >> 2][(k * NS + c3) & 3];
#pragma unroll
          for (int j = 0; j < V; ++j) {
            const float b = bv[u][k].get(j);
#pragma unroll
            for (int c3 = 0; c3 < NS; ++c3) acc[c3][j] = fmaf(sv[c3], b, acc[c3][j]);
            if constexpr (BIAS) acc[NS][j] += b;
          }
        }
      }
    }
  }
  // the quad lanes of a wave that own the same channel vector, then the waves in wave order (fixed summation order)
  for (int o = cv; o < 64; o <<= 1) {
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int j = 0; j < V; ++j) acc[r][j] += __shfl_xor(acc[r][j], o, 64);
  }
  float* slot = sacc + (1 + (threadIdx.x >> 6)) * R * cb;
  if ((int)(threadIdx.x & 63) < cv) {
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int j = 0; j < V; ++j) slot[r * cb + v * V + j] = acc[r][j];
  }
  __syncthreads();
  const int nw = blockDim.x >> 6;
  for (int i = threadIdx.x; i < R * cb; i += blockDim.x) {
    float t = 0.f;
    for (int wv = 0; wv < nw; ++wv) t += sacc[(1 + wv) * R * cb + i];
    const int r = i / cb, c = i - r * cb;
    if (r < NS) atomicAdd(out + (int64_t)r * os + (int64_t)c * oc, t);
    else atomicAdd(gbias + c, t);
  }
}

// gw[s*os + c*oc] (+)= sum over workgroups b of part[b][s][c], in workgroup order
__global__ void pw_wgrad_final_kernel(const float* __restrict__ part, int nb, int ns, int cb, int os, int oc,
                                      float* __restrict__ gw, int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ns * cb) return;
  float t = 0.f;
  for (int b = 0; b < nb; ++b) t += part[(size_t)b * ns * cb + i];
  const int s = i / cb, c = i - s * cb;
  float* dst = gw + (int64_t)s * os + (int64_t)c * oc;
  *dst = accumulate ? *dst + t : t;
}

template <typename T>
int launch_pw_fwd(const T* x, const float* w, const float* bias, T* y, int64_t npix, int cin, int cout, int wt, int epi,
                  float alpha, hipStream_t s, const T* mask = nullptr) {
  constexpr int V = Vec16<T>::N;
  const size_t lds = (size_t)(cin * cout + cout) * sizeof(float);
  if (cin <= 4) {
    if (cout % V == 0)
      hipLaunchKernelGGL((pw_small_in_kernel<T, V>), dim3(tg_grid_for(npix * (cout / V), 256)), dim3(256), lds, s, x, w,
                         bias, y, npix, cin, cout, wt, epi, alpha, mask);
    else
      hipLaunchKernelGGL((pw_small_in_kernel<T, 1>), dim3(tg_grid_for(npix * cout, 256)), dim3(256), lds, s, x, w, bias, y,
                         npix, cin, cout, wt, epi, alpha, mask);
  } else {
    if (cin % V == 0)
      hipLaunchKernelGGL((pw_small_out_kernel<T, V>), dim3(tg_grid_for(npix, 256)), dim3(256), lds, s, x, w, bias, y, npix,
                         cin, cout, wt, epi, alpha);
    else
      hipLaunchKernelGGL((pw_small_out_kernel<T, 1>), dim3(tg_grid_for(npix, 256)), dim3(256), lds, s, x, w, bias, y, npix,
                         cin, cout, wt, epi, alpha);
  }
  return TG_OK;
}

// -> 1 when the launch also added the pixel sums of gy into gbias (the RGB kernel), else 0 (gbias untouched)
template <typename T>
int launch_pw_wgrad(const T* x, const T* gy, float* gw, int64_t npix, int cin, int cout, hipStream_t s, float* gbias = nullptr) {
  constexpr int V = Vec16<T>::N;
  // small side = the <=4-channel tensor
  const bool small_in = cin <= 4;
  const T* small = small_in ? x : gy;
  const T* big = small_in ? gy : x;
  const int ns = small_in ? cin : cout, cb = small_in ? cout : cin;
  const int os = small_in ? cout : 1, oc = small_in ? 1 : cout;   // gw[ci][co] index strides
  // Every workgroup ends with ns * cb float atomics on the SAME few addresses, and with only a few trips per thread all
  // workgroups arrive there together: at 1024 workgroups the launch took ~45 us whatever the pixel count (1 M: 49.7 us,
  // 2 M: 43.9 us).  One 1024-thread workgroup per CU keeps the bytes in flight and quarters the atomics per address.
  if constexpr (sizeof(T) == 2) {
    const int cv8 = cb / 8;
    if (!exact_grid<T>() && ns == 3 && cb % 8 == 0 && cv8 <= 64 && (cv8 & (cv8 - 1)) == 0 && npix % 4 == 0 &&
        (gbias == nullptr || small_in)) {
      const int64_t nquad = npix / 4;
      const int blocks4 = tg_grid_for(nquad, 1024, 256);
      const size_t lds4 = (size_t)(3 + (gbias ? 1 : 0)) * cb * sizeof(float) * (1 + 1024 / 64);
      if (gbias)
        hipLaunchKernelGGL((pw_wgrad_rgb4_kernel<T, true>), dim3(blocks4), dim3(1024), lds4, s, small, big, gw, gbias, nquad, cb, os, oc);
      else
        hipLaunchKernelGGL((pw_wgrad_rgb4_kernel<T, false>), dim3(blocks4), dim3(1024), lds4, s, small, big, gw, gbias, nquad, cb, os, oc);
      return 1;      // the bias sum (if asked for) is done
    }
  }
  const int threads = exact_grid<T>() ? 256 : 1024;
  const size_t lds = (size_t)ns * cb * sizeof(float) * (1 + threads / 64);
  // >= 4096 pixels per workgroup: a small map (the 32 x 32 stages of the tests) is one workgroup, i.e. a fixed summation
  // order; with several workgroups the order of their atomics is the run's
  const int blocks = exact_grid<T>() ? 1 : tg_grid_for(npix, 4096, 256);
  if (cb % V == 0 && cb / V <= 256)
    hipLaunchKernelGGL((pw_wgrad_kernel<T, V>), dim3(blocks), dim3(threads), lds, s, small, big, gw, npix, ns, cb, os, oc);
  else
    hipLaunchKernelGGL((pw_wgrad_kernel<T, 1>), dim3(blocks), dim3(threads), lds, s, small, big, gw, npix, ns, cb, os, oc);
  return TG_OK;
}

// the same filter gradient from per-workgroup partials in a caller workspace, added in workgroup order
template <typename T>
int launch_pw_wgrad_ordered(const T* x, const T* gy, float* gw, int64_t npix, int cin, int cout, int accumulate, float* ws,
                            size_t ws_floats, hipStream_t s) {
  constexpr int V = Vec16<T>::N;
  const bool small_in = cin <= 4;
  const T* small = small_in ? x : gy;
  const T* big = small_in ? gy : x;
  const int ns = small_in ? cin : cout, cb = small_in ? cout : cin;
  const int os = small_in ? cout : 1, oc = small_in ? 1 : cout;
  const int threads = 1024;
  const size_t lds = (size_t)ns * cb * sizeof(float) * (1 + threads / 64);
  int blocks = tg_grid_for(npix, 4096, 256);
  if ((size_t)blocks * ns * cb > ws_floats) blocks = (int)(ws_floats / ((size_t)ns * cb));
  if (blocks < 1) return TG_EINVAL;
  if (cb % V == 0 && cb / V <= 256)
    hipLaunchKernelGGL((pw_wgrad_kernel<T, V, true>), dim3(blocks), dim3(threads), lds, s, small, big, ws, npix, ns, cb, os, oc);
  else
    hipLaunchKernelGGL((pw_wgrad_kernel<T, 1, true>), dim3(blocks), dim3(threads), lds, s, small, big, ws, npix, ns, cb, os, oc);
  hipLaunchKernelGGL(pw_wgrad_final_kernel, dim3((ns * cb + 255) / 256), dim3(256), 0, s, ws, blocks, ns, cb, os, oc, gw,
                     accumulate);
  return TG_OK;
}

}  // namespace

extern "C" {

int tg_axpby(const void* x, const void* y, void* out, int64_t numel, float a, float b, int dtype, void* stream) {
  TG_CHECK(x && out && numel >= 0, TG_EINVAL, "tg_axpby: bad arguments");
  if (numel == 0) return TG_OK;
  TG_DISPATCH_DTYPE(dtype, "tg_axpby", {
    hipLaunchKernelGGL(axpby_kernel<T>, dim3(tg_grid_for(numel / Vec16<T>::N + 1, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const T*)x, (const T*)y, (T*)out, numel, a, b);
  });
  TG_LAUNCH_CHECK("tg_axpby");
  return TG_OK;
}

int tg_rows_assemble(const TgRowsJob* jobs, int njobs, void* dst, int dtype, void* stream) {
  TG_CHECK(jobs && dst && njobs > 0 && njobs <= TG_ROWS_MAX_JOBS, TG_EINVAL, "tg_rows_assemble: 1..%d jobs", TG_ROWS_MAX_JOBS);
  RowsJobs a;
  int64_t longest = 0;
  const size_t esz = dtype == TG_F32 ? 4 : 2;
  bool vec = ((uintptr_t)dst & 15) == 0;
  for (int i = 0; i < njobs; ++i) {
    a.j[i] = jobs[i];
    const TgRowsJob& jb = jobs[i];
    TG_CHECK(jb.numel >= 0 && jb.dst_off >= 0, TG_EINVAL, "tg_rows_assemble: job %d: bad extent", i);
    bool open = true;      // sources are packed to the front
    for (int k = 0; k < 4; ++k) {
      TG_CHECK(open || !jb.src[k], TG_EINVAL, "tg_rows_assemble: job %d: source %d after an empty slot", i, k);
      open = open && jb.src[k];
      vec = vec && ((uintptr_t)jb.src[k] & 15) == 0;
    }
    vec = vec && (jb.dst_off * esz) % 16 == 0 && (jb.numel * esz) % 16 == 0;
    if (jb.numel > longest) longest = jb.numel;
  }
  if (longest == 0) return TG_OK;
  TG_DISPATCH_DTYPE(dtype, "tg_rows_assemble", {
    const int per = vec ? Vec16<T>::N * 2 : 2;      // two trips of the stride loop per thread on the longest block
    const dim3 grid(tg_grid_for((longest + per - 1) / per, 256, 2048), njobs);
    if (vec)
      hipLaunchKernelGGL((rows_assemble_kernel<T, true>), grid, dim3(256), 0, (hipStream_t)stream, a, (T*)dst);
    else
      hipLaunchKernelGGL((rows_assemble_kernel<T, false>), grid, dim3(256), 0, (hipStream_t)stream, a, (T*)dst);
  });
  TG_LAUNCH_CHECK("tg_rows_assemble");
  return TG_OK;
}

int tg_uniform(float* out, int64_t n, uint64_t seed, uint32_t* state, float lo, float hi, void* stream) {
  TG_CHECK(out && state && n > 0, TG_EINVAL, "tg_uniform: bad arguments");
  hipLaunchKernelGGL(uniform_kernel, dim3(tg_grid_for((n + 3) / 4, 256, 1024)), dim3(256), 0, (hipStream_t)stream, out, n,
                     (unsigned)seed, (unsigned)(seed >> 32), state, lo, hi - lo);
  TG_LAUNCH_CHECK("tg_uniform");
  return TG_OK;
}

int tg_lrelu_bwd(const void* gz, const void* z, void* gy, int64_t numel, float alpha, int dtype, void* stream) {
  TG_CHECK(gz && z && gy && numel >= 0, TG_EINVAL, "tg_lrelu_bwd: bad arguments");
  if (numel == 0) return TG_OK;
  TG_DISPATCH_DTYPE(dtype, "tg_lrelu_bwd", {
    hipLaunchKernelGGL(lrelu_bwd_kernel<T>, dim3(tg_grid_for(numel / Vec16<T>::N + 1, 256)), dim3(256), 0,
                       (hipStream_t)stream, (const T*)gz, (const T*)z, (T*)gy, numel, alpha);
  });
  TG_LAUNCH_CHECK("tg_lrelu_bwd");
  return TG_OK;
}

int tg_bias_lrelu_fwd(const void* y, const float* bias, void* z, int64_t npix, int c, float alpha, int dtype,
                      void* stream) {
  TG_CHECK(y && z && npix > 0 && c > 0, TG_EINVAL, "tg_bias_lrelu_fwd: bad arguments");
  const int64_t numel = npix * c;
  TG_DISPATCH_DTYPE(dtype, "tg_bias_lrelu_fwd", {
    hipLaunchKernelGGL(bias_lrelu_kernel<T>, dim3(tg_grid_for(numel / Vec16<T>::N + 1, 256)), dim3(256), 0,
                       (hipStream_t)stream, (const T*)y, bias, (T*)z, numel, c, alpha);
  });
  TG_LAUNCH_CHECK("tg_bias_lrelu_fwd");
  return TG_OK;
}

int tg_fill_scaled(void* out, const float* scalar, float value, int64_t numel, int dtype, void* stream) {
  TG_CHECK(out && numel >= 0, TG_EINVAL, "tg_fill_scaled: bad arguments");
  if (numel == 0) return TG_OK;
  TG_DISPATCH_DTYPE(dtype, "tg_fill_scaled", {
    hipLaunchKernelGGL(fill_scaled_kernel<T>, dim3(tg_grid_for(numel, 256)), dim3(256), 0, (hipStream_t)stream, (T*)out,
                       scalar, value, numel);
  });
  TG_LAUNCH_CHECK("tg_fill_scaled");
  return TG_OK;
}

int tg_cast(const void* src, void* dst, int64_t numel, int sd, int dd, void* stream) {
  TG_CHECK(src && dst && numel >= 0, TG_EINVAL, "tg_cast: bad arguments");
  if (numel == 0) return TG_OK;
  const dim3 grid(tg_grid_for(numel, 256)), blk(256);
  hipStream_t s = (hipStream_t)stream;
  if (sd == TG_F32 && dd == TG_BF16)
    hipLaunchKernelGGL((cast_kernel<float, bf16>), grid, blk, 0, s, (const float*)src, (bf16*)dst, numel);
  else if (sd == TG_BF16 && dd == TG_F32)
    hipLaunchKernelGGL((cast_kernel<bf16, float>), grid, blk, 0, s, (const bf16*)src, (float*)dst, numel);
  else if (sd == TG_F32 && dd == TG_F32)
    hipLaunchKernelGGL((cast_kernel<float, float>), grid, blk, 0, s, (const float*)src, (float*)dst, numel);
  else if (sd == TG_BF16 && dd == TG_BF16)
    hipLaunchKernelGGL((cast_kernel<bf16, bf16>), grid, blk, 0, s, (const bf16*)src, (bf16*)dst, numel);
  else if (sd == TG_F32 && dd == TG_F16)
    hipLaunchKernelGGL((cast_kernel<float, f16>), grid, blk, 0, s, (const float*)src, (f16*)dst, numel);
  else if (sd == TG_F16 && dd == TG_F32)
    hipLaunchKernelGGL((cast_kernel<f16, float>), grid, blk, 0, s, (const f16*)src, (float*)dst, numel);
  else if (sd == TG_F16 && dd == TG_F16)
    hipLaunchKernelGGL((cast_kernel<f16, f16>), grid, blk, 0, s, (const f16*)src, (f16*)dst, numel);
  else {
    tg_set_error("tg_cast: unsupported dtypes %d -> %d", sd, dd);
    return TG_EINVAL;
  }
  TG_LAUNCH_CHECK("tg_cast");
  return TG_OK;
}

int tg_sample_lerp(const void* x, const void* y, const float* alpha, void* out, int batch, int64_t per, int dtype,
                   void* stream) {
  TG_CHECK(x && y && alpha && out && batch > 0 && per > 0, TG_EINVAL, "tg_sample_lerp: bad arguments");
  TG_DISPATCH_DTYPE(dtype, "tg_sample_lerp", {
    hipLaunchKernelGGL(sample_lerp_kernel<T>, dim3(tg_grid_for(batch * per, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const T*)x, (const T*)y, alpha, (T*)out, per, batch * per);
  });
  TG_LAUNCH_CHECK("tg_sample_lerp");
  return TG_OK;
}

int tg_sample_scale(const void* x, const float* coef, const float* scalar, void* out, int batch, int64_t per, int dtype,
                    void* stream) {
  TG_CHECK(x && coef && out && batch > 0 && per > 0, TG_EINVAL, "tg_sample_scale: bad arguments");
  TG_DISPATCH_DTYPE(dtype, "tg_sample_scale", {
    hipLaunchKernelGGL(sample_scale_kernel<T>, dim3(tg_grid_for(batch * per, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const T*)x, coef, scalar, (T*)out, per, batch * per);
  });
  TG_LAUNCH_CHECK("tg_sample_scale");
  return TG_OK;
}

int tg_gdrop(const void* x, const float* noise, const float* strength_dev, float strength, int c_logical, void* out, int n,
             int64_t hw, int c, int dtype, void* stream) {
  TG_CHECK(x && noise && out && n > 0 && hw > 0 && c > 0 && c_logical > 0 && c_logical <= c, TG_EINVAL, "tg_gdrop: bad arguments");
  const int64_t per = hw * c;
  TG_DISPATCH_DTYPE(dtype, "tg_gdrop", {
    hipLaunchKernelGGL(gdrop_kernel<T>, dim3(tg_grid_for(n * per, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)x, noise,
                       strength_dev, strength, sqrtf((float)c_logical), (T*)out, per, c, n * per);
  });
  TG_LAUNCH_CHECK("tg_gdrop");
  return TG_OK;
}

#define TG_SPATIAL_LAUNCH(KERNEL, VECOK, TOTAL_VEC, TOTAL_SCALAR, ...)                                                 \
  TG_DISPATCH_DTYPE(dtype, #KERNEL, {                                                                                  \
    constexpr int V = Vec16<T>::N;                                                                                     \
    if (VECOK(V))                                                                                                      \
      hipLaunchKernelGGL((KERNEL<T, V>), dim3(tg_grid_for(TOTAL_VEC(V), 256)), dim3(256), 0, (hipStream_t)stream,      \
                         __VA_ARGS__);                                                                                 \
    else                                                                                                               \
      hipLaunchKernelGGL((KERNEL<T, 1>), dim3(tg_grid_for(TOTAL_SCALAR, 256)), dim3(256), 0, (hipStream_t)stream,      \
                         __VA_ARGS__);                                                                                 \
  })

static int check_perm(const char* who, int n, int gsz, unsigned perm, int* n1) {
  *n1 = n;
  if (!gsz) return TG_OK;
  TG_CHECK(gsz > 0 && n % gsz == 0 && n / gsz <= 4, TG_EINVAL, "%s: n (%d) must be 1..4 groups of gsz (%d)", who, n, gsz);
  int mx = 0;
  for (int k = 0; k < n / gsz; ++k) {
    const int v = (int)((perm >> (8 * k)) & 0xffu);
    if (v > mx) mx = v;
  }
  TG_CHECK(mx < 4, TG_EINVAL, "%s: permutation entry %d out of range", who, mx);
  *n1 = (mx + 1) * gsz;
  return TG_OK;
}

int tg_upsample2x_concat_fwd(const void* x0, const void* x1, void* out, int n, int h, int w, int c0, int c1, int gsz,
                             unsigned perm, int dtype, void* stream) {
  TG_CHECK(x0 && out && n > 0 && h > 0 && w > 0 && c0 > 0 && c1 >= 0 && (c1 == 0 || x1), TG_EINVAL,
           "tg_upsample2x_concat_fwd: bad arguments");
  int n1;
  int rc = check_perm("tg_upsample2x_concat_fwd", n, gsz, perm, &n1);
  if (rc) return rc;
#define VOK(V) (c0 % V == 0 && c1 % V == 0)
#define TV(V) ((int64_t)n * 4 * h * w * ((c0 + c1) / V))
  TG_SPATIAL_LAUNCH(upsample_concat_fwd_kernel, VOK, TV, (int64_t)n * 4 * h * w * (c0 + c1), (const T*)x0, (const T*)x1,
                    (T*)out, n, h, w, c0, c1, gsz, perm);
#undef VOK
#undef TV
  TG_LAUNCH_CHECK("tg_upsample2x_concat_fwd");
  return TG_OK;
}

int tg_upsample2x_concat_bwd(const void* gout, void* g0, void* g1, int n, int h, int w, int c0, int c1, int gsz,
                             unsigned perm, int dtype, void* stream) {
  TG_CHECK(gout && n > 0 && h > 0 && w > 0 && c0 > 0 && c1 >= 0, TG_EINVAL, "tg_upsample2x_concat_bwd: bad arguments");
  if (c1 == 0) g1 = nullptr;
  if (!g0 && !g1) return TG_OK;
  int n1;
  int rc = check_perm("tg_upsample2x_concat_bwd", n, gsz, perm, &n1);
  if (rc) return rc;
#define VOK(V) (c0 % V == 0 && c1 % V == 0)
#define TV(V) ((int64_t)n * h * w * (c0 / V) + (int64_t)n1 * 4 * h * w * (c1 / V))
  TG_SPATIAL_LAUNCH(upsample_concat_bwd_kernel, VOK, TV, (int64_t)n * h * w * c0 + (int64_t)n1 * 4 * h * w * c1,
                    (const T*)gout, (T*)g0, (T*)g1, n, h, w, c0, c1, gsz, perm, n1);
#undef VOK
#undef TV
  TG_LAUNCH_CHECK("tg_upsample2x_concat_bwd");
  return TG_OK;
}

int tg_pool2x2_fwd(const void* x, void* y, int n, int h, int w, int c, float scale, int dtype, void* stream) {
  TG_CHECK(x && y && n > 0 && h > 1 && w > 1 && c > 0, TG_EINVAL, "tg_pool2x2_fwd: bad arguments");
#define VOK(V) (c % V == 0)
#define TV(V) ((int64_t)n * (h / 2) * (w / 2) * (c / V))
  TG_SPATIAL_LAUNCH(pool_fwd_kernel, VOK, TV, (int64_t)n * (h / 2) * (w / 2) * c, (const T*)x, (T*)y, n, h, w, c, scale);
#undef VOK
#undef TV
  TG_LAUNCH_CHECK("tg_pool2x2_fwd");
  return TG_OK;
}

int tg_pool2x2_bwd(const void* gy, void* gx, int n, int h, int w, int c, float scale, int dtype, void* stream) {
  TG_CHECK(gy && gx && n > 0 && h > 1 && w > 1 && c > 0, TG_EINVAL, "tg_pool2x2_bwd: bad arguments");
#define VOK(V) (c % V == 0)
#define TV(V) ((int64_t)n * h * w * (c / V))
  TG_SPATIAL_LAUNCH(pool_bwd_kernel, VOK, TV, (int64_t)n * h * w * c, (const T*)gy, (T*)gx, n, h, w, c, scale);
#undef VOK
#undef TV
  TG_LAUNCH_CHECK("tg_pool2x2_bwd");
  return TG_OK;
}

int tg_pointwise_conv_fwd(const void* x, const float* w, const float* bias, void* y, int64_t npix, int cin, int cout,
                          int wt, int epilogue, float alpha, int dtype, void* stream) {
  TG_CHECK(x && w && y && npix > 0 && cin > 0 && cout > 0, TG_EINVAL, "tg_pointwise_conv_fwd: bad arguments");
  TG_CHECK(cin <= 4 || cout <= 4, TG_ENOSUP, "tg_pointwise_conv_fwd: one of cin (%d) / cout (%d) must be <= 4", cin, cout);
  TG_CHECK((size_t)(cin * cout + cout) * sizeof(float) <= 48 * 1024, TG_ENOSUP, "tg_pointwise_conv_fwd: weights too large");
  TG_CHECK(!(epilogue & TG_EPI_BIAS) || bias, TG_EINVAL, "tg_pointwise_conv_fwd: bias epilogue without bias");
  TG_DISPATCH_DTYPE(dtype, "tg_pointwise_conv_fwd", {
    launch_pw_fwd<T>((const T*)x, w, bias, (T*)y, npix, cin, cout, wt, epilogue, alpha, (hipStream_t)stream);
  });
  TG_LAUNCH_CHECK("tg_pointwise_conv_fwd");
  return TG_OK;
}

int tg_pointwise_conv_fwd_masked(const void* x, const float* w, const void* mask, void* y, int64_t npix, int cin, int cout,
                                 int wt, float alpha, int dtype, void* stream) {
  TG_CHECK(x && w && mask && y && npix > 0 && cin > 0 && cout > 0, TG_EINVAL, "tg_pointwise_conv_fwd_masked: bad arguments");
  TG_CHECK(cin <= 4, TG_ENOSUP, "tg_pointwise_conv_fwd_masked: the small side must be the input (cin = %d)", cin);
  TG_CHECK((size_t)(cin * cout + cout) * sizeof(float) <= 48 * 1024, TG_ENOSUP, "tg_pointwise_conv_fwd_masked: weights too large");
  TG_DISPATCH_DTYPE(dtype, "tg_pointwise_conv_fwd_masked", {
    launch_pw_fwd<T>((const T*)x, w, nullptr, (T*)y, npix, cin, cout, wt, 0, alpha, (hipStream_t)stream, (const T*)mask);
  });
  TG_LAUNCH_CHECK("tg_pointwise_conv_fwd_masked");
  return TG_OK;
}

int tg_pointwise_conv_bwd_weight(const void* x, const void* gy, float* gw, int64_t npix, int cin, int cout, int accumulate,
                                 int dtype, void* stream) {
  TG_CHECK(x && gy && gw && npix > 0 && cin > 0 && cout > 0, TG_EINVAL, "tg_pointwise_conv_bwd_weight: bad arguments");
  TG_CHECK(cin <= 4 || cout <= 4, TG_ENOSUP, "tg_pointwise_conv_bwd_weight: one of cin/cout must be <= 4");
  if (!accumulate) {
    int rc = tg_zero_async(gw, (size_t)cin * cout * sizeof(float), nullptr, 0, (hipStream_t)stream);
    if (rc) return rc;
  }
  TG_DISPATCH_DTYPE(dtype, "tg_pointwise_conv_bwd_weight", {
    launch_pw_wgrad<T>((const T*)x, (const T*)gy, gw, npix, cin, cout, (hipStream_t)stream);
  });
  TG_LAUNCH_CHECK("tg_pointwise_conv_bwd_weight");
  return TG_OK;
}

// The same filter gradient and, from the same read of gy, the layer's bias gradient gbias[cout] (+)= sum over pixels of gy
// (BiasAddGrad of the fromRGB layer, nets/pggan.py:233-240): one launch where the RGB kernel takes the shape, else the
// filter gradient followed by tg_channel_sum.
int tg_pointwise_conv_bwd_weight_bias(const void* x, const void* gy, float* gw, float* gbias, int64_t npix, int cin, int cout,
                                      int accumulate, int dtype, void* stream) {
  TG_CHECK(x && gy && gw && gbias && npix > 0 && cin > 0 && cout > 0, TG_EINVAL, "tg_pointwise_conv_bwd_weight_bias: bad arguments");
  TG_CHECK(cin <= 4, TG_ENOSUP, "tg_pointwise_conv_bwd_weight_bias: the small side must be the input (cin <= 4)");
  if (!accumulate) {
    int rc = tg_zero_async(gw, (size_t)cin * cout * sizeof(float), gbias, (size_t)cout * sizeof(float), (hipStream_t)stream);
    if (rc) return rc;
  }
  int fused = 0;
  TG_DISPATCH_DTYPE(dtype, "tg_pointwise_conv_bwd_weight_bias", {
    fused = launch_pw_wgrad<T>((const T*)x, (const T*)gy, gw, npix, cin, cout, (hipStream_t)stream, gbias);
  });
  TG_LAUNCH_CHECK("tg_pointwise_conv_bwd_weight_bias");
  if (!fused) return tg_channel_sum(gy, gbias, npix, cout, 1, dtype, stream);
  return TG_OK;
}

int tg_pointwise_conv_bwd_weight_ordered(const void* x, const void* gy, float* gw, int64_t npix, int cin, int cout,
                                         int accumulate, float* ws, size_t ws_floats, int dtype, void* stream) {
  TG_CHECK(x && gy && gw && ws && npix > 0 && cin > 0 && cout > 0, TG_EINVAL, "tg_pointwise_conv_bwd_weight_ordered: bad arguments");
  TG_CHECK(cin <= 4 || cout <= 4, TG_ENOSUP, "tg_pointwise_conv_bwd_weight_ordered: one of cin/cout must be <= 4");
  int rc = TG_OK;
  TG_DISPATCH_DTYPE(dtype, "tg_pointwise_conv_bwd_weight_ordered", {
    rc = launch_pw_wgrad_ordered<T>((const T*)x, (const T*)gy, gw, npix, cin, cout, accumulate, ws, ws_floats,
                                    (hipStream_t)stream);
  });
  TG_CHECK(rc == TG_OK, TG_EINVAL, "tg_pointwise_conv_bwd_weight_ordered: workspace of %zu floats is too small", ws_floats);
  TG_LAUNCH_CHECK("tg_pointwise_conv_bwd_weight_ordered");
  return TG_OK;
}

}  // extern "C"
