// Minibatch-stddev (forward, backward, double backward), loss reductions, WGAN-GP tail, the small
// dense layer and the fused Adam step.
#include "tg_common.h"

namespace {

// The n (<= 16) samples of V consecutive positions, all requested before any is used: a runtime-length load/use
// loop is one L2 round trip per sample (3 such loops made these single-workgroup kernels 20-90 us).
constexpr int MB_NMAX = 16;
// A thread's POSITION vector is 8 bytes (4 positions of a 16-bit type, 2 of fp32), not 16: the [16, 4, 4, 256] tensor then
// spreads over 1024 threads instead of 512 -- these kernels' time is the per-thread chain (16 samples x V conversions,
// sums, a division or two per position), not bytes; with 16-byte vectors half of the forward's 1024 threads idled and the
// backward kernels needed 226-256 VGPRs.
template <typename T, int V>
struct alignas(sizeof(T) * V) MbVec {
  T e[V];
  __device__ __forceinline__ float get(int i) const { return (float)e[i]; }
  __device__ __forceinline__ void set(int i, float x) { e[i] = (T)x; }
};
template <typename T> constexpr int mb_v() { return 8 / (int)sizeof(T); }
template <typename T, int V> __device__ __forceinline__ MbVec<T, V> mb_ld(const T* p) {
  return *reinterpret_cast<const MbVec<T, V>*>(p);
}
template <typename T, int V> __device__ __forceinline__ void mb_st(T* p, const MbVec<T, V>& v) {
  *reinterpret_cast<MbVec<T, V>*>(p) = v;
}
template <typename T, int V>
__device__ __forceinline__ void mb_load_samples(const T* __restrict__ x, int n, int P, int p, MbVec<T, V> (&xv)[MB_NMAX]) {
#pragma unroll
  for (int i = 0; i < MB_NMAX; ++i) xv[i] = mb_ld<T, V>(x + (int64_t)(i < n ? i : n - 1) * P + p);
}


// out[px][0..c) = src[px][0..c), out[px][c] = val, out[px][c+1..cpad) = 0 over nvec 16-byte vectors of the padded tensor,
// U vectors per thread in flight (a load -> store loop of one vector per trip is one L2 round trip per trip; U = 9 with
// 1024 threads takes the [16, 4, 4, 264] tensor in ONE trip)
template <typename T, int U = 4>
__device__ __forceinline__ void mb_copy_with_stat(const T* __restrict__ src, T* __restrict__ out, int64_t nvec, int c,
                                                  int cpad, float val) {
  constexpr int V = Vec16<T>::N;
  const int cvp = cpad / V;
  for (int64_t i0 = threadIdx.x; i0 < nvec; i0 += (int64_t)blockDim.x * U) {
    Vec16<T> o[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int64_t i = i0 + (int64_t)u * blockDim.x;
      i = i < nvec ? i : nvec - 1;
      const int cb = (int)(i % cvp) * V;
      const int64_t px = i / cvp;
      if (cb < c) {
        o[u] = ldv(src + px * c + cb);
      } else {
#pragma unroll
        for (int j = 0; j < V; ++j) o[u].set(j, (cb + j == c) ? val : 0.f);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + (int64_t)u * blockDim.x;
      if (i < nvec) stv(out + (i / cvp) * cpad + (int)(i % cvp) * V, o[u]);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Minibatch stddev, nets/pggan_utils.py:353-366.  x[n][p], p = hw*c.  Single workgroup: the tensor
// is [B,4,4,C] (64 K elements at B=16, C=256).
//   stat = mean_p sqrt(var_n(x[:,p]) + eps)          (biased variance over the batch)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(1024) void mbstd_fwd_kernel(const T* __restrict__ x, T* __restrict__ out,
                                                         float* __restrict__ stat, int n, int hw, int c, int cpad,
                                                         float eps) {
  __shared__ float red[16];
  const int P = hw * c;
  // one workgroup per statistic group of n images (blockIdx.x): the reference computes the statistic per
  // discriminator call; several calls batched along N keep their own statistic
  x += (int64_t)blockIdx.x * n * P;
  out += (int64_t)blockIdx.x * n * hw * cpad;
  if (stat) stat += blockIdx.x;
  float acc = 0.f;
  constexpr int V = mb_v<T>();
  if (P % V == 0 && n <= 32) {
    // a thread owns V consecutive positions and keeps the n samples' vectors in flight together
    for (int p = threadIdx.x * V; p < P; p += blockDim.x * V) {
      float mu[V], var[V];
#pragma unroll
      for (int j = 0; j < V; ++j) mu[j] = var[j] = 0.f;
      if (n <= MB_NMAX) {
        MbVec<T, V> xs[MB_NMAX];
        mb_load_samples<T, V>(x, n, P, p, xs);
#pragma unroll
        for (int i = 0; i < MB_NMAX; ++i)
          if (i < n) {
#pragma unroll
            for (int j = 0; j < V; ++j) mu[j] += xs[i].get(j);
          }
#pragma unroll
        for (int j = 0; j < V; ++j) mu[j] /= (float)n;
#pragma unroll
        for (int i = 0; i < MB_NMAX; ++i)
          if (i < n) {
#pragma unroll
            for (int j = 0; j < V; ++j) {
              const float d = xs[i].get(j) - mu[j];
              var[j] = fmaf(d, d, var[j]);
            }
          }
      } else {
        for (int i = 0; i < n; ++i) {
          const MbVec<T, V> xv = mb_ld<T, V>(x + (int64_t)i * P + p);
#pragma unroll
          for (int j = 0; j < V; ++j) mu[j] += xv.get(j);
        }
#pragma unroll
        for (int j = 0; j < V; ++j) mu[j] /= (float)n;
        for (int i = 0; i < n; ++i) {
          const MbVec<T, V> xv = mb_ld<T, V>(x + (int64_t)i * P + p);
#pragma unroll
          for (int j = 0; j < V; ++j) {
            const float d = xv.get(j) - mu[j];
            var[j] = fmaf(d, d, var[j]);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < V; ++j) acc += sqrtf(var[j] / (float)n + eps);
    }
  } else {
    for (int p = threadIdx.x; p < P; p += blockDim.x) {
      float mu = 0.f;
      for (int i = 0; i < n; ++i) mu += ld(x + (int64_t)i * P + p);
      mu /= (float)n;
      float var = 0.f;
      for (int i = 0; i < n; ++i) {
        const float d = ld(x + (int64_t)i * P + p) - mu;
        var = fmaf(d, d, var);
      }
      acc += sqrtf(var / (float)n + eps);
    }
  }
  const float val = block_sum(acc, red) / (float)P;
  if (threadIdx.x == 0 && stat) stat[0] = val;
  const int64_t total = (int64_t)n * hw * cpad;
  constexpr int V16 = Vec16<T>::N;
  if (c % V16 == 0 && cpad % V16 == 0) {
    mb_copy_with_stat<T, 9>(x, out, total / V16, c, cpad, val);
  } else {
    for (int64_t i = threadIdx.x; i < total; i += blockDim.x) {
      const int ch = (int)(i % cpad);
      const int64_t px = i / cpad;
      float v = 0.f;
      if (ch < c)
        v = ld(x + px * c + ch);
      else if (ch == c)
        v = val;
      st(out + i, v);
    }
  }
}

// gx[n][p] = gout[n][hw][ch<c] + G * (x - mu_p) / (N * sigma_p * P),  G = sum over (n,hw) of gout[..., c]
// 256-thread workgroups, gridDim.y of them per statistic group, one 8-byte position vector per thread and trip (four
// workgroups cover the [16, 4, 4, 256] bf16 tensor; each sums G for itself).  History: the single 1024-thread workgroup
// with 16-byte vectors spilled 168-392 bytes per thread to scratch and took 24-52 us for 400 KB of traffic
// (profiles/r03_z_shapes_eager_step.json); two 256-thread workgroups with 16-byte vectors 256 VGPRs and 15 us.
// The first trip's loads are requested BEFORE the reduction of G, whose barrier they do not depend on.
template <typename T>
__global__ __launch_bounds__(256) void mbstd_bwd_kernel(const T* __restrict__ gout, const T* __restrict__ x,
                                                        T* __restrict__ gx, int n, int hw, int c, int cpad, float eps) {
  __shared__ float red[16];
  const int P = hw * c;
  gout += (int64_t)blockIdx.x * n * hw * cpad;
  x += (int64_t)blockIdx.x * n * P;
  gx += (int64_t)blockIdx.x * n * P;
  constexpr int V = mb_v<T>();
  const bool fast = c % V == 0 && cpad % V == 0 && n <= MB_NMAX;
  MbVec<T, V> xs[MB_NMAX], gs[MB_NMAX];
  const int pstep = gridDim.y * blockDim.x * V;
  const int pfirst = (blockIdx.y * blockDim.x + threadIdx.x) * V;
  if (fast) {      // dead threads (pfirst >= P) read position 0: harmless, never used
    const int p = pfirst < P ? pfirst : 0;
    const int px = p / c, ch = p - px * c;
    mb_load_samples<T, V>(x, n, P, p, xs);
#pragma unroll
    for (int i = 0; i < MB_NMAX; ++i) gs[i] = mb_ld<T, V>(gout + ((int64_t)(i < n ? i : n - 1) * hw + px) * cpad + ch);
  }
  float acc = 0.f;
  for (int i = threadIdx.x; i < n * hw; i += blockDim.x) acc += ld(gout + (int64_t)i * cpad + c);
  const float G = block_sum(acc, red);
  if (fast) {
    for (int p = pfirst; p < P; p += pstep) {
      float mu[V], var[V], k[V];
      if (p != pfirst) {
        const int px = p / c, ch = p - px * c;
        mb_load_samples<T, V>(x, n, P, p, xs);
#pragma unroll
        for (int i = 0; i < MB_NMAX; ++i) gs[i] = mb_ld<T, V>(gout + ((int64_t)(i < n ? i : n - 1) * hw + px) * cpad + ch);
      }
#pragma unroll
      for (int j = 0; j < V; ++j) mu[j] = var[j] = 0.f;
#pragma unroll
      for (int i = 0; i < MB_NMAX; ++i)
        if (i < n) {
#pragma unroll
          for (int j = 0; j < V; ++j) mu[j] += xs[i].get(j);
        }
#pragma unroll
      for (int j = 0; j < V; ++j) mu[j] /= (float)n;
#pragma unroll
      for (int i = 0; i < MB_NMAX; ++i)
        if (i < n) {
#pragma unroll
          for (int j = 0; j < V; ++j) {
            const float d = xs[i].get(j) - mu[j];
            var[j] = fmaf(d, d, var[j]);
          }
        }
#pragma unroll
      for (int j = 0; j < V; ++j) k[j] = G / ((float)n * sqrtf(var[j] / (float)n + eps) * (float)P);
#pragma unroll
      for (int i = 0; i < MB_NMAX; ++i)
        if (i < n) {
          MbVec<T, V> o;
#pragma unroll
          for (int j = 0; j < V; ++j) o.set(j, gs[i].get(j) + k[j] * (xs[i].get(j) - mu[j]));
          mb_st<T, V>(gx + (int64_t)i * P + p, o);
        }
    }
    return;
  }
  if (c % V == 0 && cpad % V == 0 && n <= 32) {
    for (int p = pfirst; p < P; p += pstep) {
      float mu[V], var[V], k[V];
#pragma unroll
      for (int j = 0; j < V; ++j) mu[j] = var[j] = 0.f;
      for (int i = 0; i < n; ++i) {
        const MbVec<T, V> xv = mb_ld<T, V>(x + (int64_t)i * P + p);
#pragma unroll
        for (int j = 0; j < V; ++j) mu[j] += xv.get(j);
      }
#pragma unroll
      for (int j = 0; j < V; ++j) mu[j] /= (float)n;
      for (int i = 0; i < n; ++i) {
        const MbVec<T, V> xv = mb_ld<T, V>(x + (int64_t)i * P + p);
#pragma unroll
        for (int j = 0; j < V; ++j) {
          const float d = xv.get(j) - mu[j];
          var[j] = fmaf(d, d, var[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < V; ++j) k[j] = G / ((float)n * sqrtf(var[j] / (float)n + eps) * (float)P);
      const int px = p / c, ch = p - px * c;
      for (int i = 0; i < n; ++i) {
        const MbVec<T, V> xv = mb_ld<T, V>(x + (int64_t)i * P + p);
        const MbVec<T, V> gv = mb_ld<T, V>(gout + ((int64_t)i * hw + px) * cpad + ch);
        MbVec<T, V> o;
#pragma unroll
        for (int j = 0; j < V; ++j) o.set(j, gv.get(j) + k[j] * (xv.get(j) - mu[j]));
        mb_st<T, V>(gx + (int64_t)i * P + p, o);
      }
    }
    return;
  }
  for (int p = blockIdx.y * blockDim.x + threadIdx.x; p < P; p += gridDim.y * blockDim.x) {
    float mu = 0.f;
    for (int i = 0; i < n; ++i) mu += ld(x + (int64_t)i * P + p);
    mu /= (float)n;
    float var = 0.f;
    for (int i = 0; i < n; ++i) {
      const float d = ld(x + (int64_t)i * P + p) - mu;
      var = fmaf(d, d, var);
    }
    const float sigma = sqrtf(var / (float)n + eps);
    const float k = G / ((float)n * sigma * (float)P);
    const int px = p / c, ch = p - px * c;
    for (int i = 0; i < n; ++i) {
      const float d = ld(x + (int64_t)i * P + p) - mu;
      st(gx + (int64_t)i * P + p, ld(gout + ((int64_t)i * hw + px) * cpad + ch) + k * d);
    }
  }
}

// Double backward.  With c_n = x_n - mu, sigma = sqrt(mean c^2 + eps), gx_n = gpass_n + G c_n /(N sigma P):
//   d/dG      : T = sum_{n,p} v_np c_np / (N sigma_p P)  -> ggout[..., c] = T for every (n,hw); ggout[..., <c] = v
//   d/dx_mp   : (G/(N P)) * [ (v_m - mean_n v)/sigma - (sum_n v_n c_n) c_m / (N sigma^3) ]
// 1024 threads, 8-byte position vectors (one trip at [16, 4, 4, 256]), first trip's loads ahead of the reduction of G (see
// mbstd_bwd_kernel); the per-element divisions by sigma / sigma^3 are multiplications by per-POSITION reciprocals (two
// fp32 divisions per element were most of this kernel's instructions: 35 us for 400 KB)
template <typename T, int NTHR>      // NTHR: 1024 for the 16-bit types, 512 for fp32 (its 128-VGPR budget at 1024 spilled)
__global__ __launch_bounds__(NTHR) void mbstd_bwd_bwd_kernel(const T* __restrict__ v, const T* __restrict__ gout,
                                                            const T* __restrict__ x, T* __restrict__ ggout,
                                                            T* __restrict__ gx2, int n, int hw, int c, int cpad,
                                                            float eps) {
  __shared__ float red[16];
  const int P = hw * c;
  v += (int64_t)blockIdx.x * n * P;
  gout += (int64_t)blockIdx.x * n * hw * cpad;
  x += (int64_t)blockIdx.x * n * P;
  if (ggout) ggout += (int64_t)blockIdx.x * n * hw * cpad;
  if (gx2) gx2 += (int64_t)blockIdx.x * n * P;
  constexpr int V = mb_v<T>();
  const bool fast = P % V == 0 && n <= MB_NMAX;
  MbVec<T, V> xs[MB_NMAX], vs[MB_NMAX];
  const int pfirst = threadIdx.x * V;
  if (fast) {      // dead threads (pfirst >= P) read position 0: harmless, never used
    mb_load_samples<T, V>(x, n, P, pfirst < P ? pfirst : 0, xs);
    mb_load_samples<T, V>(v, n, P, pfirst < P ? pfirst : 0, vs);
  }
  float acc = 0.f;
  for (int i = threadIdx.x; i < n * hw; i += blockDim.x) acc += ld(gout + (int64_t)i * cpad + c);
  const float G = block_sum(acc, red);
  float tacc = 0.f;
  if (fast) {
    for (int p = pfirst; p < P; p += blockDim.x * V) {
      if (p != pfirst) {
        mb_load_samples<T, V>(x, n, P, p, xs);
        mb_load_samples<T, V>(v, n, P, p, vs);
      }
      float mu[V], vm[V], var[V], vc[V], sigma[V];
#pragma unroll
      for (int j = 0; j < V; ++j) mu[j] = vm[j] = var[j] = vc[j] = 0.f;
#pragma unroll
      for (int i = 0; i < MB_NMAX; ++i)
        if (i < n) {
#pragma unroll
          for (int j = 0; j < V; ++j) {
            mu[j] += xs[i].get(j);
            vm[j] += vs[i].get(j);
          }
        }
#pragma unroll
      for (int j = 0; j < V; ++j) {
        mu[j] /= (float)n;
        vm[j] /= (float)n;
      }
#pragma unroll
      for (int i = 0; i < MB_NMAX; ++i)
        if (i < n) {
#pragma unroll
          for (int j = 0; j < V; ++j) {
            const float d = xs[i].get(j) - mu[j];
            var[j] = fmaf(d, d, var[j]);
            vc[j] = fmaf(vs[i].get(j), d, vc[j]);
          }
        }
#pragma unroll
      for (int j = 0; j < V; ++j) {
        sigma[j] = sqrtf(var[j] / (float)n + eps);
        tacc += vc[j] / ((float)n * sigma[j] * (float)P);
      }
      if (gx2) {
        const float k = G / ((float)n * (float)P);
        float is[V], cf[V];      // 1 / sigma,  sum_n(v c) / (N sigma^3)
#pragma unroll
        for (int j = 0; j < V; ++j) {
          is[j] = 1.f / sigma[j];
          cf[j] = vc[j] * is[j] * is[j] * is[j] / (float)n;
        }
#pragma unroll
        for (int i = 0; i < MB_NMAX; ++i)
          if (i < n) {
            MbVec<T, V> o;
#pragma unroll
            for (int j = 0; j < V; ++j) {
              const float d = xs[i].get(j) - mu[j];
              o.set(j, k * ((vs[i].get(j) - vm[j]) * is[j] - cf[j] * d));
            }
            mb_st<T, V>(gx2 + (int64_t)i * P + p, o);
          }
      }
    }
  }
  for (int p = threadIdx.x; p < P && !fast; p += blockDim.x) {
    float mu = 0.f, vm = 0.f;
    for (int i = 0; i < n; ++i) {
      mu += ld(x + (int64_t)i * P + p);
      vm += ld(v + (int64_t)i * P + p);
    }
    mu /= (float)n;
    vm /= (float)n;
    float var = 0.f, vc = 0.f;
    for (int i = 0; i < n; ++i) {
      const float d = ld(x + (int64_t)i * P + p) - mu;
      var = fmaf(d, d, var);
      vc = fmaf(ld(v + (int64_t)i * P + p), d, vc);
    }
    const float sigma = sqrtf(var / (float)n + eps);
    tacc += vc / ((float)n * sigma * (float)P);
    if (gx2) {
      const float k = G / ((float)n * (float)P);
      for (int i = 0; i < n; ++i) {
        const float d = ld(x + (int64_t)i * P + p) - mu;
        const float vi = ld(v + (int64_t)i * P + p);
        st(gx2 + (int64_t)i * P + p, k * ((vi - vm) / sigma - vc * d / ((float)n * sigma * sigma * sigma)));
      }
    }
  }
  const float Tt = block_sum(tacc, red);
  if (ggout) {
    const int64_t total = (int64_t)n * hw * cpad;
    constexpr int V16 = Vec16<T>::N;
    if (c % V16 == 0 && cpad % V16 == 0) {
      mb_copy_with_stat<T, 9>(v, ggout, total / V16, c, cpad, Tt);
    } else {
      for (int64_t i = threadIdx.x; i < total; i += blockDim.x) {
        const int ch = (int)(i % cpad);
        const int64_t px = i / cpad;
        float o = 0.f;
        if (ch < c)
          o = ld(v + px * c + ch);
        else if (ch == c)
          o = Tt;
        st(ggout + i, o);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// reductions
// ------------------------------------------------------------------------------------------------
// MODE 0: sum(x)  1: sum|x-y|;  PART: out[blockIdx.x] = this workgroup's sum;  direct (a ONE-workgroup launch that does not
// accumulate): out[0] is written, not added to -- the launch needs no zero fill before it
template <typename T, int MODE, bool PART = false>
__global__ void sum_kernel(const T* __restrict__ x, const T* __restrict__ y, float* __restrict__ out, int64_t numel,
                           float scale, int direct = 0) {
  __shared__ float red[8];
  constexpr int V = Vec16<T>::N;
  const int64_t nvec = numel / V, stride = (int64_t)gridDim.x * blockDim.x;
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    Vec16<T> a = ldv(x + i * V);
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < V; ++j) acc += a.get(j);
    } else {
      Vec16<T> b = ldv(y + i * V);
#pragma unroll
      for (int j = 0; j < V; ++j) acc += fabsf(a.get(j) - b.get(j));
    }
  }
  for (int64_t i = nvec * V + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += stride)
    acc += MODE == 0 ? ld(x + i) : fabsf(ld(x + i) - ld(y + i));
  const float tot = block_sum(acc, red);
  if (threadIdx.x == 0) {
    if constexpr (PART) out[blockIdx.x] = tot;
    else if (direct) out[0] = tot * scale;
    else atomicAdd(out, tot * scale);
  }
}

// out[0] (+)= scale * (part[0] + part[1] + ... in index order)
__global__ void ordered_scalar_sum_kernel(const float* __restrict__ part, int n, float* __restrict__ out, float scale,
                                          int accumulate) {
  if (blockIdx.x || threadIdx.x) return;
  float t = 0.f;
  for (int i = 0; i < n; ++i) t += part[i];
  out[0] = accumulate ? out[0] + t * scale : t * scale;
}

template <typename T>
__global__ void abs_diff_bwd_kernel(const T* __restrict__ a, const T* __restrict__ b, const float* __restrict__ gscale,
                                    T* __restrict__ ga, T* __restrict__ gb, int64_t numel, float scale) {
  const float g = (gscale ? gscale[0] : 1.f) * scale;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (int64_t)gridDim.x * blockDim.x) {
    const float d = ld(a + i) - ld(b + i);
    const float s = d > 0.f ? g : (d < 0.f ? -g : 0.f);
    if (ga) st(ga + i, s);
    if (gb) st(gb + i, -s);
  }
}

// out[b] = sum over sample b of x^2;  grid = (chunks, batch), out pre-zeroed
template <typename T>
__global__ void sample_sumsq_kernel(const T* __restrict__ x, float* __restrict__ out, int64_t per) {
  __shared__ float red[8];
  constexpr int V = Vec16<T>::N;
  const T* base = x + (int64_t)blockIdx.y * per;
  const int64_t nvec = per / V, stride = (int64_t)gridDim.x * blockDim.x;
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    Vec16<T> a = ldv(base + i * V);
#pragma unroll
    for (int j = 0; j < V; ++j) acc = fmaf(a.get(j), a.get(j), acc);
  }
  for (int64_t i = nvec * V + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += stride) {
    const float a = ld(base + i);
    acc = fmaf(a, a, acc);
  }
  const float tot = block_sum(acc, red);
  if (threadIdx.x == 0) atomicAdd(out + blockIdx.y, tot);
}

// image_generation.py:431-436
__global__ void gp_penalty_kernel(const float* __restrict__ ss, float* __restrict__ loss, float* __restrict__ coef,
                                  int batch, float lambda) {
  __shared__ float red[8];
  float acc = 0.f;
  for (int b = threadIdx.x; b < batch; b += blockDim.x) {
    const float slope = sqrtf(ss[b]);
    const float d = slope - 1.f;
    acc += d * d;
    if (coef) coef[b] = slope > 0.f ? lambda * 2.f * d / (slope * (float)batch) : 0.f;
  }
  const float tot = block_sum(acc, red);
  if (threadIdx.x == 0 && loss) loss[0] = lambda * tot / (float)batch;
}

// Prediction losses on the tiny fp32 [B,1] discriminator outputs (image_generation.py:331-400).
//   mode 0: x                      (WGAN means)
//   mode 1: relu(a + b*x)          (hinge: a = 1, b = +1 for fakes / -1 for reals)
//   mode 2: sigmoid cross entropy against label a in {0,1}: max(x,0) - x*a + log(1 + exp(-|x|))
//   mode 3: x^2                    (WGAN drift term)
__device__ __forceinline__ float pred_loss_f(float x, int mode, float a, float b) {
  if (mode == 1) return fmaxf(a + b * x, 0.f);
  if (mode == 2) return fmaxf(x, 0.f) - x * a + log1pf(expf(-fabsf(x)));
  if (mode == 3) return x * x;
  return x;
}
__device__ __forceinline__ float pred_loss_df(float x, int mode, float a, float b) {
  if (mode == 1) return (a + b * x) > 0.f ? b : 0.f;
  if (mode == 2) return 1.f / (1.f + expf(-x)) - a;       // sigmoid(x) - label
  if (mode == 3) return 2.f * x;
  return 1.f;
}
__global__ void pred_loss_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int n, int mode, float a,
                                     float b, float scale, int accumulate) {
  __shared__ float red[8];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) acc += pred_loss_f(x[i], mode, a, b);
  const float tot = block_sum(acc, red);
  if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + scale * tot;
}
__global__ void pred_loss_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gscale, float* __restrict__ gx,
                                     int n, int mode, float a, float b, float scale) {
  const float g = (gscale ? gscale[0] : 1.f) * scale;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    gx[i] = g * pred_loss_df(x[i], mode, a, b);
}

// ---- the loss tail of one batched discriminator call in ONE launch each way (image_generation.py:331-400).  pred holds
// `groups` predictions of group_size rows ([real; cyc; prime] ...); job j adds coef * mean_i f_mode(x_i; a, b) over group
// `group` to term `term`.  Jobs and gradient pointers travel by value in the kernel arguments (a hipGraph records them).
constexpr int PRED_MAX_JOBS = 12, PRED_MAX_TERMS = 8;
struct PredJobs {
  TgPredJob j[PRED_MAX_JOBS];
  int n;
};
struct PredGrads {
  const float* g[PRED_MAX_TERMS];      // d loss / d term t: a device fp32 scalar each (NULL: that term has no gradient)
};
__global__ __launch_bounds__(256) void pred_losses_fwd_kernel(const float* __restrict__ x, int gs, PredJobs jobs, float* __restrict__ terms,
                                                              int nterms) {
  __shared__ float red[8];
  __shared__ float acc_t[PRED_MAX_TERMS];
  if (threadIdx.x < PRED_MAX_TERMS) acc_t[threadIdx.x] = 0.f;
  __syncthreads();
  for (int j = 0; j < jobs.n; ++j) {      // in job order: a fixed summation order
    const TgPredJob J = jobs.j[j];
    float acc = 0.f;
    for (int i = threadIdx.x; i < gs; i += blockDim.x) acc += pred_loss_f(x[J.group * gs + i], J.mode, J.a, J.b);
    const float tot = block_sum(acc, red);
    if (threadIdx.x == 0) acc_t[J.term] += J.coef * tot / (float)gs;
    __syncthreads();
  }
  if (threadIdx.x < nterms) terms[threadIdx.x] = acc_t[threadIdx.x];
}
__global__ void pred_losses_bwd_kernel(const float* __restrict__ x, int gs, int groups, PredJobs jobs, PredGrads gr,
                                       float* __restrict__ gx) {
  const int n = gs * groups;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int grp = i / gs;
    const float xi = x[i];
    float g = 0.f;
    for (int j = 0; j < jobs.n; ++j) {
      const TgPredJob J = jobs.j[j];
      if (J.group == grp && gr.g[J.term]) g += gr.g[J.term][0] * (J.coef / (float)gs) * pred_loss_df(xi, J.mode, J.a, J.b);
    }
    gx[i] = g;
  }
}
// out[0] = sum_i x_i[0] over up to 24 device fp32 scalars (tf.add_n over the loss collection), in argument order
struct ScalarPtrs {
  const float* p[24];
  int n;
};
__global__ void sum_scalars_kernel(ScalarPtrs a, float* __restrict__ out) {
  if (blockIdx.x || threadIdx.x) return;
  float t = 0.f;
  for (int i = 0; i < a.n; ++i) t += a.p[i][0];
  out[0] = t;
}

// out[0] = E[x^2] - E[x]^2 from sum = s1[0] and the per-sample sums of squares ss[batch]  (DRAGAN,
// image_generation.py:445: the VARIANCE over every element of the minibatch)
// tf.losses.cosine_distance(l2n(expected), l2n(embedding), axis=-1, weights=w) (twingan.py:515-519):
//   out = (w / B) * sum_b (1 - e_b . p_b / (|e_b| |p_b|)),  norms clamped as tf.nn.l2_normalize does (sum x^2 >= 1e-12).
// One workgroup walks the rows in order: a handful of [B, D] rows, fixed summation order.
__global__ __launch_bounds__(256) void cosine_distance_fwd_kernel(const float* __restrict__ e, const float* __restrict__ p,
                                                                  float* __restrict__ out, int b, int d, float scale) {
  __shared__ float red[4];
  float tot = 0.f;
  for (int r = 0; r < b; ++r) {
    float dot = 0.f, se = 0.f, sp = 0.f;
    for (int c = threadIdx.x; c < d; c += 256) {
      const float x = e[(size_t)r * d + c], y = p[(size_t)r * d + c];
      dot = fmaf(x, y, dot);
      se = fmaf(x, x, se);
      sp = fmaf(y, y, sp);
    }
    dot = block_sum(dot, red);
    se = block_sum(se, red);
    sp = block_sum(sp, red);
    tot += 1.f - dot * rsqrtf(fmaxf(se, 1e-12f)) * rsqrtf(fmaxf(sp, 1e-12f));
  }
  if (threadIdx.x == 0) out[0] = tot * scale;
}

// d out / d p[b, c] = -(w / B) * g * (ehat_c - phat_c (ehat . phat)) / |p|   (|p|^2 >= 1e-12; below it phat = p * 1e6)
__global__ __launch_bounds__(256) void cosine_distance_bwd_kernel(const float* __restrict__ e, const float* __restrict__ p,
                                                                  const float* __restrict__ gscale, float* __restrict__ gp,
                                                                  int d, float scale) {
  __shared__ float red[4];
  const int r = blockIdx.x;
  float dot = 0.f, se = 0.f, sp = 0.f;
  for (int c = threadIdx.x; c < d; c += 256) {
    const float x = e[(size_t)r * d + c], y = p[(size_t)r * d + c];
    dot = fmaf(x, y, dot);
    se = fmaf(x, x, se);
    sp = fmaf(y, y, sp);
  }
  dot = block_sum(dot, red);
  se = block_sum(se, red);
  sp = block_sum(sp, red);
  const float ie = rsqrtf(fmaxf(se, 1e-12f)), ip = rsqrtf(fmaxf(sp, 1e-12f));
  const float cosv = dot * ie * ip, k = -scale * gscale[0];
  const bool clamped = sp < 1e-12f;
  for (int c = threadIdx.x; c < d; c += 256) {
    const float eh = e[(size_t)r * d + c] * ie, ph = p[(size_t)r * d + c] * ip;
    gp[(size_t)r * d + c] = k * (clamped ? eh * ip : (eh - ph * cosv) * ip);
  }
}

__global__ void var_from_sums_kernel(const float* __restrict__ s1, const float* __restrict__ ss, float* __restrict__ out,
                                     int batch, float inv_numel) {
  __shared__ float red[8];
  float acc = 0.f;
  for (int i = threadIdx.x; i < batch; i += blockDim.x) acc += ss[i];
  const float tot = block_sum(acc, red);
  if (threadIdx.x == 0) {
    const float m = s1[0] * inv_numel;
    out[0] = fmaxf(tot * inv_numel - m * m, 0.f);
  }
}

// ---- the fully connected layer of a network's tail (layers.fully_connected, nets/pggan_utils.py:323-327) as ONE launch each
// way: y[B,N] = x[B,K] @ w[K,N] + b[N] with x in the activations' storage type (the cast, the GEMM and the bias add were
// three launches), and gx = g @ w^T (in x's type), gw (+)= x^T @ g, gb (+)= sum_b g (four launches).  B <= 64-ish rows,
// K = 256, N = 1 (the discriminators' prediction) .. 16: one WAVE per output element forward (lanes split K), one THREAD
// per k backward (it owns row k of gw and column k of gx).
template <typename T>
__global__ void fc_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                              float* __restrict__ y, int m, int n, int k) {
  const int64_t total = (int64_t)m * n;
  const int lane = threadIdx.x & 63;
  const int64_t w0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t i = w0; i < total; i += nw) {
    const int col = (int)(i % n), row = (int)(i / n);
    float acc = 0.f;
    for (int kk = lane; kk < k; kk += 64) acc = fmaf(ld(x + (int64_t)row * k + kk), w[(int64_t)kk * n + col], acc);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) y[i] = acc + (b ? b[col] : 0.f);
  }
}
template <typename T>
__global__ void fc_bwd_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ g,
                              T* __restrict__ gx, float* __restrict__ gw, float* __restrict__ gb, int m, int n, int k,
                              int acc_w, int acc_b) {
  const int kk = blockIdx.x * blockDim.x + threadIdx.x;
  if (kk < k) {
    for (int row = 0; gx && row < m; ++row) {      // gx[row, kk] = sum_col g[row, col] * w[kk, col]
      float a = 0.f;
      for (int col = 0; col < n; ++col) a = fmaf(g[(int64_t)row * n + col], w[(int64_t)kk * n + col], a);
      st(gx + (int64_t)row * k + kk, a);
    }
    for (int col = 0; gw && col < n; ++col) {      // gw[kk, col] (+)= sum_row x[row, kk] * g[row, col], rows in order
      float a = 0.f;
      for (int row = 0; row < m; ++row) a = fmaf(ld(x + (int64_t)row * k + kk), g[(int64_t)row * n + col], a);
      gw[(int64_t)kk * n + col] = (acc_w ? gw[(int64_t)kk * n + col] : 0.f) + a;
    }
  }
  if (gb && kk < n) {                              // gb[col] (+)= sum_row g[row, col]
    float a = 0.f;
    for (int row = 0; row < m; ++row) a += g[(int64_t)row * n + kk];
    gb[kk] = (acc_b ? gb[kk] : 0.f) + a;
  }
}

// C[m,n] (+)= op(A)[m,k] @ op(B)[k,n] + bias[n]; one thread per output element
__global__ void small_gemm_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ bias,
                                  float* __restrict__ c, int m, int n, int k, int ta, int tb, int accumulate) {
  const int64_t total = (int64_t)m * n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int col = (int)(i % n), row = (int)(i / n);
    float acc = 0.f;
    for (int kk = 0; kk < k; ++kk) {
      const float av = ta ? a[(int64_t)kk * m + row] : a[(int64_t)row * k + kk];
      const float bv = tb ? b[(int64_t)col * k + kk] : b[(int64_t)kk * n + col];
      acc = fmaf(av, bv, acc);
    }
    if (bias) acc += bias[col];
    c[i] = (accumulate ? c[i] : 0.f) + acc;
  }
}

// Same GEMM with one WAVE per output element (lanes split K, butterfly sum): the discriminator's fully connected
// layer is [B,256] x [256,1] -- a thread per output walks K = 256 as one dependent chain (18 us).
__global__ void small_gemm_wave_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                       const float* __restrict__ bias, float* __restrict__ c, int m, int n, int k, int ta,
                                       int tb, int accumulate) {
  const int64_t total = (int64_t)m * n;
  const int lane = threadIdx.x & 63;
  const int64_t w0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t i = w0; i < total; i += nw) {
    const int col = (int)(i % n), row = (int)(i / n);
    float acc = 0.f;
    for (int kk = lane; kk < k; kk += 64) {
      const float av = ta ? a[(int64_t)kk * m + row] : a[(int64_t)row * k + kk];
      const float bv = tb ? b[(int64_t)col * k + kk] : b[(int64_t)kk * n + col];
      acc = fmaf(av, bv, acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) {
      if (bias) acc += bias[col];
      c[i] = (accumulate ? c[i] : 0.f) + acc;
    }
  }
}

// TF-1.x Adam (model/model_inheritor.py:537-542): epsilon OUTSIDE the bias-corrected sqrt.
__global__ void adam_kernel(float* __restrict__ th, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, bf16* __restrict__ shadow, int64_t numel, float lr_t,
                            const float* __restrict__ lr_t_dev, float b1, float b2, float eps, float gscale) {
  if (lr_t_dev) lr_t = lr_t_dev[0];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * gscale;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    const float t = th[i] - lr_t * mi / (sqrtf(vi) + eps);
    m[i] = mi;
    v[i] = vi;
    th[i] = t;
    if (shadow) shadow[i] = (bf16)t;
  }
}

// One thread: t = ++step; lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t)   (tf.train.AdamOptimizer._prepare/_apply_dense)
__global__ void adam_tick_kernel(int64_t* step, float* lr_t, float lr, float b1, float b2) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const int64_t t = step[0] + 1;
    step[0] = t;
    lr_t[0] = (float)((double)lr * sqrt(1.0 - pow((double)b2, (double)t)) / (1.0 - pow((double)b1, (double)t)));
  }
}

}  // namespace

extern "C" {

int tg_mbstd_fwd(const void* x, void* out, float* stat, int n, int groups, int hw, int c, int cpad, float eps, int dtype,
                 void* stream) {
  TG_CHECK(x && out && n > 0 && hw > 0 && c > 0 && cpad > c, TG_EINVAL, "tg_mbstd_fwd: bad arguments (cpad must exceed c)");
  TG_CHECK(groups > 0 && n % groups == 0, TG_EINVAL, "tg_mbstd_fwd: n (%d) not divisible by groups (%d)", n, groups);
  TG_DISPATCH_DTYPE(dtype, "tg_mbstd_fwd", {
    hipLaunchKernelGGL(mbstd_fwd_kernel<T>, dim3(groups), dim3(1024), 0, (hipStream_t)stream, (const T*)x, (T*)out, stat,
                       n / groups, hw, c, cpad, eps);
  });
  TG_LAUNCH_CHECK("tg_mbstd_fwd");
  return TG_OK;
}

int tg_mbstd_bwd(const void* gout, const void* x, void* gx, int n, int groups, int hw, int c, int cpad, float eps,
                 int dtype, void* stream) {
  TG_CHECK(gout && x && gx && n > 0 && hw > 0 && c > 0 && cpad > c, TG_EINVAL, "tg_mbstd_bwd: bad arguments");
  TG_CHECK(groups > 0 && n % groups == 0, TG_EINVAL, "tg_mbstd_bwd: n (%d) not divisible by groups (%d)", n, groups);
  TG_DISPATCH_DTYPE(dtype, "tg_mbstd_bwd", {
    int nsplit = (hw * c / mb_v<T>() + 255) / 256;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > 8) nsplit = 8;
    hipLaunchKernelGGL(mbstd_bwd_kernel<T>, dim3(groups, nsplit), dim3(256), 0, (hipStream_t)stream, (const T*)gout, (const T*)x,
                       (T*)gx, n / groups, hw, c, cpad, eps);
  });
  TG_LAUNCH_CHECK("tg_mbstd_bwd");
  return TG_OK;
}

int tg_mbstd_bwd_bwd(const void* v, const void* gout, const void* x, void* ggout, void* gx2, int n, int groups, int hw, int c,
                     int cpad, float eps, int dtype, void* stream) {
  TG_CHECK(v && gout && x && n > 0 && hw > 0 && c > 0 && cpad > c, TG_EINVAL, "tg_mbstd_bwd_bwd: bad arguments");
  TG_CHECK(groups > 0 && n % groups == 0, TG_EINVAL, "tg_mbstd_bwd_bwd: n (%d) not divisible by groups (%d)", n, groups);
  TG_DISPATCH_DTYPE(dtype, "tg_mbstd_bwd_bwd", {
    constexpr int NTHR = sizeof(T) == 4 ? 512 : 1024;
    hipLaunchKernelGGL((mbstd_bwd_bwd_kernel<T, NTHR>), dim3(groups), dim3(NTHR), 0, (hipStream_t)stream, (const T*)v,
                       (const T*)gout, (const T*)x, (T*)ggout, (T*)gx2, n / groups, hw, c, cpad, eps);
  });
  TG_LAUNCH_CHECK("tg_mbstd_bwd_bwd");
  return TG_OK;
}

static int zero_unless(float* p, size_t bytes, int accumulate, hipStream_t s, const char* who) {
  if (!accumulate) return tg_zero_async(p, bytes, nullptr, 0, s);
  return TG_OK;
}

int tg_sum(const void* x, float* out, int64_t numel, float scale, int accumulate, int dtype, void* stream) {
  TG_CHECK(x && out && numel > 0, TG_EINVAL, "tg_sum: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  TG_DISPATCH_DTYPE(dtype, "tg_sum", {
    const int blocks = exact_grid<T>() ? 1 : tg_grid_for(numel / Vec16<T>::N + 1, 256, 1024);
    const int direct = (blocks == 1 && !accumulate) ? 1 : 0;      // the [B, 1] predictions of the loss tails: one launch, not two
    if (!direct) {
      int rc = zero_unless(out, sizeof(float), accumulate, s, "tg_sum");
      if (rc) return rc;
    }
    hipLaunchKernelGGL((sum_kernel<T, 0>), dim3(blocks), dim3(256), 0, s, (const T*)x, (const T*)nullptr, out, numel, scale, direct);
  });
  TG_LAUNCH_CHECK("tg_sum");
  return TG_OK;
}

int tg_abs_diff_sum(const void* a, const void* b, float* out, int64_t numel, float scale, int accumulate, int dtype,
                    void* stream) {
  TG_CHECK(a && b && out && numel > 0, TG_EINVAL, "tg_abs_diff_sum: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  TG_DISPATCH_DTYPE(dtype, "tg_abs_diff_sum", {
    const int blocks = exact_grid<T>() ? 1 : tg_grid_for(numel / Vec16<T>::N + 1, 256, 1024);
    const int direct = (blocks == 1 && !accumulate) ? 1 : 0;
    if (!direct) {
      int rc = zero_unless(out, sizeof(float), accumulate, s, "tg_abs_diff_sum");
      if (rc) return rc;
    }
    hipLaunchKernelGGL((sum_kernel<T, 1>), dim3(blocks), dim3(256), 0, s, (const T*)a, (const T*)b, out, numel, scale, direct);
  });
  TG_LAUNCH_CHECK("tg_abs_diff_sum");
  return TG_OK;
}

// The two sums above with per-workgroup partials in a caller workspace, added in workgroup order: the same value on
// every run whatever the storage type and whatever the mode (the plain entry points end in one fp32 atomic per workgroup)
int tg_sum_ordered(const void* x, const void* y_or_null, float* out, int64_t numel, float scale, int accumulate, float* ws,
                   size_t ws_floats, int dtype, void* stream) {
  TG_CHECK(x && out && ws && ws_floats >= 1 && numel > 0, TG_EINVAL, "tg_sum_ordered: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  int blocks = 0;
  TG_DISPATCH_DTYPE(dtype, "tg_sum_ordered", {
    blocks = tg_grid_for(numel / Vec16<T>::N + 1, 256, 512);
    if ((size_t)blocks > ws_floats) blocks = (int)ws_floats;
    if (y_or_null)
      hipLaunchKernelGGL((sum_kernel<T, 1, true>), dim3(blocks), dim3(256), 0, s, (const T*)x, (const T*)y_or_null, ws, numel, 1.f);
    else
      hipLaunchKernelGGL((sum_kernel<T, 0, true>), dim3(blocks), dim3(256), 0, s, (const T*)x, (const T*)nullptr, ws, numel, 1.f);
  });
  hipLaunchKernelGGL(ordered_scalar_sum_kernel, dim3(1), dim3(64), 0, s, ws, blocks, out, scale, accumulate);
  TG_LAUNCH_CHECK("tg_sum_ordered");
  return TG_OK;
}

int tg_abs_diff_bwd(const void* a, const void* b, const float* gscale, void* ga, void* gb, int64_t numel, float scale,
                    int dtype, void* stream) {
  TG_CHECK(a && b && numel > 0 && (ga || gb), TG_EINVAL, "tg_abs_diff_bwd: bad arguments");
  TG_DISPATCH_DTYPE(dtype, "tg_abs_diff_bwd", {
    hipLaunchKernelGGL(abs_diff_bwd_kernel<T>, dim3(tg_grid_for(numel, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const T*)a, (const T*)b, gscale, (T*)ga, (T*)gb, numel, scale);
  });
  TG_LAUNCH_CHECK("tg_abs_diff_bwd");
  return TG_OK;
}

int tg_sample_sumsq(const void* x, float* out, int batch, int64_t per, int dtype, void* stream) {
  TG_CHECK(x && out && batch > 0 && per > 0, TG_EINVAL, "tg_sample_sumsq: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  int rc = zero_unless(out, (size_t)batch * sizeof(float), 0, s, "tg_sample_sumsq");
  if (rc) return rc;
  TG_DISPATCH_DTYPE(dtype, "tg_sample_sumsq", {
    const int chunks = exact_grid<T>() ? 1 : tg_grid_for(per / Vec16<T>::N + 1, 256, 64);
    hipLaunchKernelGGL(sample_sumsq_kernel<T>, dim3(chunks, batch), dim3(256), 0, s, (const T*)x, out, per);
  });
  TG_LAUNCH_CHECK("tg_sample_sumsq");
  return TG_OK;
}

int tg_gp_penalty(const float* sumsq, float* loss, float* coef, int batch, float lambda, void* stream) {
  TG_CHECK(sumsq && batch > 0, TG_EINVAL, "tg_gp_penalty: bad arguments");
  hipLaunchKernelGGL(gp_penalty_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, sumsq, loss, coef, batch, lambda);
  TG_LAUNCH_CHECK("tg_gp_penalty");
  return TG_OK;
}

int tg_pred_loss_fwd(const float* x, float* out, int n, int mode, float a, float b, float scale, int accumulate,
                     void* stream) {
  TG_CHECK(x && out && n > 0 && mode >= 0 && mode <= 3, TG_EINVAL, "tg_pred_loss_fwd: bad arguments");
  hipLaunchKernelGGL(pred_loss_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, out, n, mode, a, b, scale,
                     accumulate);
  TG_LAUNCH_CHECK("tg_pred_loss_fwd");
  return TG_OK;
}

int tg_pred_loss_bwd(const float* x, const float* gscale, float* gx, int n, int mode, float a, float b, float scale,
                     void* stream) {
  TG_CHECK(x && gx && n > 0 && mode >= 0 && mode <= 3, TG_EINVAL, "tg_pred_loss_bwd: bad arguments");
  hipLaunchKernelGGL(pred_loss_bwd_kernel, dim3(tg_grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, gscale, gx, n,
                     mode, a, b, scale);
  TG_LAUNCH_CHECK("tg_pred_loss_bwd");
  return TG_OK;
}

int tg_pred_losses_fwd(const float* pred, int group_size, int groups, const TgPredJob* jobs, int njobs, float* terms, int nterms,
                       void* stream) {
  TG_CHECK(pred && jobs && terms && group_size > 0 && groups > 0 && njobs > 0 && njobs <= PRED_MAX_JOBS && nterms > 0 &&
               nterms <= PRED_MAX_TERMS, TG_EINVAL, "tg_pred_losses_fwd: bad arguments (at most %d jobs, %d terms)", PRED_MAX_JOBS,
           PRED_MAX_TERMS);
  PredJobs pj;
  pj.n = njobs;
  for (int j = 0; j < njobs; ++j) {
    TG_CHECK(jobs[j].group >= 0 && jobs[j].group < groups && jobs[j].term >= 0 && jobs[j].term < nterms && jobs[j].mode >= 0 &&
                 jobs[j].mode <= 3, TG_EINVAL, "tg_pred_losses_fwd: job %d out of range", j);
    pj.j[j] = jobs[j];
  }
  hipLaunchKernelGGL(pred_losses_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, pred, group_size, pj, terms, nterms);
  TG_LAUNCH_CHECK("tg_pred_losses_fwd");
  return TG_OK;
}

int tg_pred_losses_bwd(const float* pred, int group_size, int groups, const TgPredJob* jobs, int njobs, const float* const* gterms,
                       int nterms, float* gpred, void* stream) {
  TG_CHECK(pred && jobs && gterms && gpred && group_size > 0 && groups > 0 && njobs > 0 && njobs <= PRED_MAX_JOBS && nterms > 0 &&
               nterms <= PRED_MAX_TERMS, TG_EINVAL, "tg_pred_losses_bwd: bad arguments");
  PredJobs pj;
  pj.n = njobs;
  for (int j = 0; j < njobs; ++j) {
    TG_CHECK(jobs[j].group >= 0 && jobs[j].group < groups && jobs[j].term >= 0 && jobs[j].term < nterms, TG_EINVAL,
             "tg_pred_losses_bwd: job %d out of range", j);
    pj.j[j] = jobs[j];
  }
  PredGrads pg;
  for (int t = 0; t < PRED_MAX_TERMS; ++t) pg.g[t] = t < nterms ? gterms[t] : nullptr;
  hipLaunchKernelGGL(pred_losses_bwd_kernel, dim3(tg_grid_for((int64_t)group_size * groups, 256, 64)), dim3(256), 0,
                     (hipStream_t)stream, pred, group_size, groups, pj, pg, gpred);
  TG_LAUNCH_CHECK("tg_pred_losses_bwd");
  return TG_OK;
}

int tg_sum_scalars(const float* const* scalars, int n, float* out, void* stream) {
  TG_CHECK(scalars && out && n > 0 && n <= 24, TG_EINVAL, "tg_sum_scalars: 1..24 scalars");
  ScalarPtrs a;
  a.n = n;
  for (int i = 0; i < n; ++i) {
    TG_CHECK(scalars[i], TG_EINVAL, "tg_sum_scalars: null scalar %d", i);
    a.p[i] = scalars[i];
  }
  hipLaunchKernelGGL(sum_scalars_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, out);
  TG_LAUNCH_CHECK("tg_sum_scalars");
  return TG_OK;
}

int tg_cosine_distance_fwd(const float* expected, const float* embedding, float* out, int batch, int dim, float weight,
                           void* stream) {
  TG_CHECK(expected && embedding && out && batch > 0 && dim > 0, TG_EINVAL, "tg_cosine_distance_fwd: bad arguments");
  hipLaunchKernelGGL(cosine_distance_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, expected, embedding, out, batch,
                     dim, weight / (float)batch);
  TG_LAUNCH_CHECK("tg_cosine_distance_fwd");
  return TG_OK;
}

int tg_cosine_distance_bwd(const float* expected, const float* embedding, const float* gscale, float* g_embedding, int batch,
                           int dim, float weight, void* stream) {
  TG_CHECK(expected && embedding && gscale && g_embedding && batch > 0 && dim > 0, TG_EINVAL,
           "tg_cosine_distance_bwd: bad arguments");
  hipLaunchKernelGGL(cosine_distance_bwd_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, expected, embedding, gscale,
                     g_embedding, dim, weight / (float)batch);
  TG_LAUNCH_CHECK("tg_cosine_distance_bwd");
  return TG_OK;
}

int tg_var_from_sums(const float* sum, const float* sample_sumsq, float* out, int batch, int64_t numel, void* stream) {
  TG_CHECK(sum && sample_sumsq && out && batch > 0 && numel > 0, TG_EINVAL, "tg_var_from_sums: bad arguments");
  hipLaunchKernelGGL(var_from_sums_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, sum, sample_sumsq, out, batch,
                     1.0f / (float)numel);
  TG_LAUNCH_CHECK("tg_var_from_sums");
  return TG_OK;
}

int tg_small_gemm(const float* a, const float* b, const float* bias, float* c, int m, int n, int k, int ta, int tb,
                  int accumulate, void* stream) {
  TG_CHECK(a && b && c && m > 0 && n > 0 && k > 0, TG_EINVAL, "tg_small_gemm: bad arguments");
  if (k >= 32 && (int64_t)m * n <= 4096)
    hipLaunchKernelGGL(small_gemm_wave_kernel, dim3(tg_grid_for((int64_t)m * n * 64, 256)), dim3(256), 0,
                       (hipStream_t)stream, a, b, bias, c, m, n, k, ta, tb, accumulate);
  else
    hipLaunchKernelGGL(small_gemm_kernel, dim3(tg_grid_for((int64_t)m * n, 256)), dim3(256), 0, (hipStream_t)stream, a, b,
                       bias, c, m, n, k, ta, tb, accumulate);
  TG_LAUNCH_CHECK("tg_small_gemm");
  return TG_OK;
}

int tg_fc_fwd(const void* x, const float* w, const float* bias, float* y, int m, int n, int k, int dtype, void* stream) {
  TG_CHECK(x && w && y && m > 0 && n > 0 && k > 0, TG_EINVAL, "tg_fc_fwd: bad arguments");
  TG_DISPATCH_DTYPE(dtype, "tg_fc_fwd", {
    hipLaunchKernelGGL(fc_fwd_kernel<T>, dim3(tg_grid_for((int64_t)m * n * 64, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const T*)x, w, bias, y, m, n, k);
  });
  TG_LAUNCH_CHECK("tg_fc_fwd");
  return TG_OK;
}

int tg_fc_bwd(const void* x, const float* w, const float* g, void* gx, float* gw, float* gb, int m, int n, int k, int acc_w,
              int acc_b, int dtype, void* stream) {
  TG_CHECK(x && w && g && (gx || gw || gb) && m > 0 && n > 0 && k > 0 && n <= k, TG_EINVAL, "tg_fc_bwd: bad arguments");
  TG_DISPATCH_DTYPE(dtype, "tg_fc_bwd", {
    hipLaunchKernelGGL(fc_bwd_kernel<T>, dim3(tg_grid_for(k, 64)), dim3(64), 0, (hipStream_t)stream, (const T*)x, w, g, (T*)gx,
                       gw, gb, m, n, k, acc_w, acc_b);
  });
  TG_LAUNCH_CHECK("tg_fc_bwd");
  return TG_OK;
}

int tg_adam_step(float* theta, const float* grad, float* m, float* v, void* theta_bf16, int64_t numel, float lr_t,
                 const float* lr_t_dev, float beta1, float beta2, float eps, float grad_scale, void* stream) {
  TG_CHECK(theta && grad && m && v && numel > 0, TG_EINVAL, "tg_adam_step: bad arguments");
  hipLaunchKernelGGL(adam_kernel, dim3(tg_grid_for(numel, 256)), dim3(256), 0, (hipStream_t)stream, theta, grad, m, v,
                     (bf16*)theta_bf16, numel, lr_t, lr_t_dev, beta1, beta2, eps, grad_scale);
  TG_LAUNCH_CHECK("tg_adam_step");
  return TG_OK;
}

int tg_adam_tick(int64_t* step_dev, float* lr_t_dev, float lr, float beta1, float beta2, void* stream) {
  TG_CHECK(step_dev && lr_t_dev, TG_EINVAL, "tg_adam_tick: null pointer");
  hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step_dev, lr_t_dev, lr, beta1, beta2);
  TG_LAUNCH_CHECK("tg_adam_tick");
  return TG_OK;
}

}  // extern "C"
