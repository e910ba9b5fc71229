// Spectral normalisation of a conv kernel (libs/sn.py:38-101, called from nets/pggan_utils.py:316-320 under
// --spectral_norm): one power iteration from the persistent vector u,
//     v = l2n(u W^T),  u' = l2n(v W),  sigma = v W u'^T,  W_bar = W / sigma          (W = the kernel as [K = k*k*cin, cout])
// and its backward with the gradient flowing through sigma, v and u' (the reference stops nothing):
//     d sigma / d W = v (x) u' + b (x) u,   b = (a - v (v . a)) / |u W^T|,  a = W u'^T
//     d L / d W = G / sigma - (sum(G o W) / sigma^2) * d sigma / d W                (G = d L / d W_bar)
// (u'^T d u' = 0 kills the path through u'; checked against autograd in tests/test_oracle.py.)
// fp32 master weights; every sum is a two-stage reduction in a fixed order (no atomics).
#include "tg_common.h"

namespace {

constexpr int ROWS = 16;      // rows of W per block of the row-dot kernels
constexpr int KS = 16;        // K splits of the column-dot kernel

__device__ __forceinline__ float sum_parts(const float* part, int n, float* red) {
  float t = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) t += part[i];
  return block_sum(t, red);
}

// out[r] = sum_c W[r,c] vec[c];  part_ss[blk] = sum over the block's rows of out[r]^2;
// WITH_G: part_s[blk] = sum over the block's rows of sum_c G[r,c] W[r,c], part_va[blk] = sum_r v[r] out[r]
template <bool WITH_G>
__global__ __launch_bounds__(256) void sn_rowdot(const float* __restrict__ w, const float* __restrict__ vec,
                                                 float* __restrict__ out, float* __restrict__ part_ss,
                                                 const float* __restrict__ g, const float* __restrict__ v,
                                                 float* __restrict__ part_s, float* __restrict__ part_va, int k_rows,
                                                 int cout) {
  __shared__ float red[3][4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float ss = 0.f, s = 0.f, va = 0.f;
  for (int j = 0; j < ROWS / 4; ++j) {
    const int r = blockIdx.x * ROWS + wid * (ROWS / 4) + j;
    if (r >= k_rows) break;      // wave-uniform
    float d = 0.f, gs = 0.f;
    for (int c = lane; c < cout; c += 64) {
      const float x = w[(size_t)r * cout + c];
      d = fmaf(x, vec[c], d);
      if (WITH_G) gs = fmaf(x, g[(size_t)r * cout + c], gs);
    }
    d = wave_sum(d);
    if (WITH_G) gs = wave_sum(gs);
    if (lane == 0) out[r] = d;
    ss = fmaf(d, d, ss);
    if (WITH_G) {
      s += gs;
      va = fmaf(v[r], d, va);
    }
  }
  if (lane == 0) {
    red[0][wid] = ss;
    red[1][wid] = s;
    red[2][wid] = va;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (!WITH_G) part_ss[blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    if (WITH_G) {
      part_s[blockIdx.x] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
      part_va[blockIdx.x] = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
    }
  }
}

// upart[ks][c] = (1 / |v_raw|) * sum over rows of K split ks of v_raw[k] W[k,c].   grid = (ceil(cout / 64), KS)
__global__ __launch_bounds__(256) void sn_coldot(const float* __restrict__ w, const float* __restrict__ v_raw,
                                                 const float* __restrict__ part_ss, int nb1, float* __restrict__ upart,
                                                 int k_rows, int cout) {
  __shared__ float red[4];
  __shared__ float acc[4][64];
  const float ssv = sum_parts(part_ss, nb1, red);
  const float inv = rsqrtf(fmaxf(ssv, 1e-12f));      // tf.nn.l2_normalize: x * rsqrt(max(sum x^2, 1e-12))
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rs = threadIdx.x >> 6;
  const int per = (k_rows + KS - 1) / KS;
  const int k0 = blockIdx.y * per, k1 = min(k0 + per, k_rows);
  float a = 0.f;
  if (c < cout)
    for (int k = k0 + rs; k < k1; k += 4) a = fmaf(v_raw[k], w[(size_t)k * cout + c], a);
  acc[rs][threadIdx.x & 63] = a;
  __syncthreads();
  if (rs == 0 && c < cout) {
    const int l = threadIdx.x;
    upart[(size_t)blockIdx.y * cout + c] = ((acc[0][l] + acc[1][l]) + (acc[2][l] + acc[3][l])) * inv;
  }
}

// sigma, u' from the column partials (every block, redundantly: KS * cout floats); w_bar = W / sigma for the block's
// elements; block 0 writes u_new, v (normalised) and stats = {sigma, |v_raw|}.   cout <= 1024
__global__ __launch_bounds__(256) void sn_finish(const float* __restrict__ w, const float* __restrict__ upart,
                                                 const float* __restrict__ v_raw, const float* __restrict__ part_ss, int nb1,
                                                 float* __restrict__ w_bar, float* __restrict__ u_new, float* __restrict__ v,
                                                 float* __restrict__ stats, int k_rows, int cout) {
  __shared__ float red[4];
  float ur[4], ssu = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = threadIdx.x + j * 256;
    float t = 0.f;
    if (c < cout)
      for (int ks = 0; ks < KS; ++ks) t += upart[(size_t)ks * cout + c];
    ur[j] = t;
    ssu = fmaf(t, t, ssu);
  }
  ssu = block_sum(ssu, red);
  const float inv_u = rsqrtf(fmaxf(ssu, 1e-12f));
  const float sigma = ssu * inv_u;                     // v W u'^T = u_raw . u' = |u_raw|^2 / max(|u_raw|, 1e-6)
  const float inv_sigma = 1.f / sigma;
  const size_t total = (size_t)k_rows * cout;
  for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < min(total, (size_t)(blockIdx.x + 1) * 1024); i += 256)
    w_bar[i] = w[i] * inv_sigma;
  if (blockIdx.x == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = threadIdx.x + j * 256;
      if (c < cout) u_new[c] = ur[j] * inv_u;
    }
    const float ssv = sum_parts(part_ss, nb1, red);
    const float inv_v = rsqrtf(fmaxf(ssv, 1e-12f));
    for (int k = threadIdx.x; k < k_rows; k += 256) v[k] = v_raw[k] * inv_v;
    if (threadIdx.x == 0) {
      stats[0] = sigma;
      stats[1] = 1.f / inv_v;
    }
  }
}

// gw (+)= G / sigma - (s / sigma^2) * (v[k] u'[c] + b[k] u[c]),  b[k] = (a[k] - v[k] va) / |v_raw|
__global__ __launch_bounds__(256) void sn_bwd_apply(const float* __restrict__ g, const float* __restrict__ u,
                                                    const float* __restrict__ u_new, const float* __restrict__ v,
                                                    const float* __restrict__ a, const float* __restrict__ part_s,
                                                    const float* __restrict__ part_va, int nb1,
                                                    const float* __restrict__ stats, float* __restrict__ gw, int accumulate,
                                                    int k_rows, int cout) {
  __shared__ float red[4];
  const float s = sum_parts(part_s, nb1, red);
  const float va = sum_parts(part_va, nb1, red);
  const float sigma = stats[0], nv = stats[1];
  const float inv_sigma = 1.f / sigma, coef = s * inv_sigma * inv_sigma, inv_nv = 1.f / nv;
  const size_t total = (size_t)k_rows * cout;
  for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < min(total, (size_t)(blockIdx.x + 1) * 1024); i += 256) {
    const int k = (int)(i / cout), c = (int)(i - (size_t)k * cout);
    const float b = (a[k] - v[k] * va) * inv_nv;
    const float r = g[i] * inv_sigma - coef * (v[k] * u_new[c] + b * u[c]);
    gw[i] = accumulate ? gw[i] + r : r;
  }
}

inline int nb_rows(int k_rows) { return (k_rows + ROWS - 1) / ROWS; }

// ---- the three forward kernels' bodies once more, with the block index as an argument, for the *_multi kernels below
// (copies rather than a refactoring of the kernels above: those stay byte-identical to the build the GPU suite ran on)
template <bool WITH_G>
__device__ __forceinline__ void sn_rowdot_body(const int bx, const float* __restrict__ w, const float* __restrict__ vec,
                                               float* __restrict__ out, float* __restrict__ part_ss,
                                               const float* __restrict__ g, const float* __restrict__ v,
                                               float* __restrict__ part_s, float* __restrict__ part_va, int k_rows, int cout) {
  __shared__ float red[3][4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float ss = 0.f, s = 0.f, va = 0.f;
  for (int j = 0; j < ROWS / 4; ++j) {
    const int r = bx * ROWS + wid * (ROWS / 4) + j;
    if (r >= k_rows) break;      // wave-uniform
    float d = 0.f, gs = 0.f;
    for (int c = lane; c < cout; c += 64) {
      const float x = w[(size_t)r * cout + c];
      d = fmaf(x, vec[c], d);
      if (WITH_G) gs = fmaf(x, g[(size_t)r * cout + c], gs);
    }
    d = wave_sum(d);
    if (WITH_G) gs = wave_sum(gs);
    if (lane == 0) out[r] = d;
    ss = fmaf(d, d, ss);
    if (WITH_G) {
      s += gs;
      va = fmaf(v[r], d, va);
    }
  }
  if (lane == 0) {
    red[0][wid] = ss;
    red[1][wid] = s;
    red[2][wid] = va;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (!WITH_G) part_ss[bx] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    if (WITH_G) {
      part_s[bx] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
      part_va[bx] = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
    }
  }
}

__device__ __forceinline__ void sn_coldot_body(const int bx, const int by, const float* __restrict__ w,
                                               const float* __restrict__ v_raw, const float* __restrict__ part_ss, int nb1,
                                               float* __restrict__ upart, int k_rows, int cout) {
  __shared__ float red[4];
  __shared__ float acc[4][64];
  const float ssv = sum_parts(part_ss, nb1, red);
  const float inv = rsqrtf(fmaxf(ssv, 1e-12f));      // tf.nn.l2_normalize: x * rsqrt(max(sum x^2, 1e-12))
  const int c = bx * 64 + (threadIdx.x & 63), rs = threadIdx.x >> 6;
  const int per = (k_rows + KS - 1) / KS;
  const int k0 = by * per, k1 = min(k0 + per, k_rows);
  float a = 0.f;
  if (c < cout)
    for (int k = k0 + rs; k < k1; k += 4) a = fmaf(v_raw[k], w[(size_t)k * cout + c], a);
  acc[rs][threadIdx.x & 63] = a;
  __syncthreads();
  if (rs == 0 && c < cout) {
    const int l = threadIdx.x;
    upart[(size_t)by * cout + c] = ((acc[0][l] + acc[1][l]) + (acc[2][l] + acc[3][l])) * inv;
  }
}

__device__ __forceinline__ void sn_finish_body(const int bx, const float* __restrict__ w, const float* __restrict__ upart,
                                               const float* __restrict__ v_raw, const float* __restrict__ part_ss, int nb1,
                                               float* __restrict__ w_bar, float* __restrict__ u_new, float* __restrict__ v,
                                               float* __restrict__ stats, int k_rows, int cout) {
  __shared__ float red[4];
  float ur[4], ssu = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = threadIdx.x + j * 256;
    float t = 0.f;
    if (c < cout)
      for (int ks = 0; ks < KS; ++ks) t += upart[(size_t)ks * cout + c];
    ur[j] = t;
    ssu = fmaf(t, t, ssu);
  }
  ssu = block_sum(ssu, red);
  const float inv_u = rsqrtf(fmaxf(ssu, 1e-12f));
  const float sigma = ssu * inv_u;                     // v W u'^T = u_raw . u' = |u_raw|^2 / max(|u_raw|, 1e-6)
  const float inv_sigma = 1.f / sigma;
  const size_t total = (size_t)k_rows * cout;
  for (size_t i = (size_t)bx * 1024 + threadIdx.x; i < min(total, (size_t)(bx + 1) * 1024); i += 256)
    w_bar[i] = w[i] * inv_sigma;
  if (bx == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = threadIdx.x + j * 256;
      if (c < cout) u_new[c] = ur[j] * inv_u;
    }
    const float ssv = sum_parts(part_ss, nb1, red);
    const float inv_v = rsqrtf(fmaxf(ssv, 1e-12f));
    for (int k = threadIdx.x; k < k_rows; k += 256) v[k] = v_raw[k] * inv_v;
    if (threadIdx.x == 0) {
      stats[0] = sigma;
      stats[1] = 1.f / inv_v;
    }
  }
}

// One power iteration of MANY kernels in three launches (tg_spectral_norm_fwd_multi): a job table in device memory, the
// blocks of each of the three passes laid end to end (row0 / col0 / fin0 = a job's first block in that pass).
struct SnJob {
  const float* w;
  const float* u;
  float* w_bar;
  float* u_new;
  float* v;
  float* stats;
  float* ws;      // tg_spectral_norm_workspace(k_rows, cout) bytes
  int k_rows, cout, nb;
  int row0, col0, fin0;
};
template <int WHICH>
__device__ __forceinline__ int sn_job_of(const SnJob* __restrict__ t, int njobs, int blk) {
  int j = 0;
  for (int i = 1; i < njobs; ++i) {
    const int first = WHICH == 0 ? t[i].row0 : WHICH == 1 ? t[i].col0 : t[i].fin0;
    if (first <= blk) j = i;
  }
  return j;
}
// u <- u_new for every job (libs/sn.py:84-86: the assign of the power-iteration vector at the end of a run), one workgroup
// per job: 30 device-to-device copies of a few hundred bytes in BASELINE configs[4] otherwise
__global__ __launch_bounds__(256) void sn_assign_u_multi(const SnJob* __restrict__ t) {
  const SnJob J = t[blockIdx.x];
  float* u = const_cast<float*>(J.u);
  for (int i = threadIdx.x; i < J.cout; i += 256) u[i] = J.u_new[i];
}
__global__ __launch_bounds__(256) void sn_rowdot_multi(const SnJob* __restrict__ t, int njobs) {
  const SnJob J = t[sn_job_of<0>(t, njobs, blockIdx.x)];
  float* v_raw = J.ws;
  sn_rowdot_body<false>(blockIdx.x - J.row0, J.w, J.u, v_raw, v_raw + J.k_rows, nullptr, nullptr, nullptr, nullptr, J.k_rows,
                        J.cout);
}
__global__ __launch_bounds__(256) void sn_coldot_multi(const SnJob* __restrict__ t, int njobs) {
  const SnJob J = t[sn_job_of<1>(t, njobs, blockIdx.x)];
  float* v_raw = J.ws;
  float* part_ss = v_raw + J.k_rows;
  const int local = blockIdx.x - J.col0, ncb = (J.cout + 63) / 64;
  sn_coldot_body(local % ncb, local / ncb, J.w, v_raw, part_ss, J.nb, part_ss + 3 * J.nb, J.k_rows, J.cout);
}
__global__ __launch_bounds__(256) void sn_finish_multi(const SnJob* __restrict__ t, int njobs) {
  const SnJob J = t[sn_job_of<2>(t, njobs, blockIdx.x)];
  float* v_raw = J.ws;
  float* part_ss = v_raw + J.k_rows;
  sn_finish_body(blockIdx.x - J.fin0, J.w, part_ss + 3 * J.nb, v_raw, part_ss, J.nb, J.w_bar, J.u_new, J.v, J.stats, J.k_rows,
                 J.cout);
}


}  // namespace

extern "C" {

// floats: v_raw / a [K] | part_ss [nb] | part_s [nb] | part_va [nb] | upart [KS * cout]
size_t tg_spectral_norm_workspace(int k_rows, int cout) {
  if (k_rows <= 0 || cout <= 0) return 0;
  return sizeof(float) * ((size_t)k_rows + 3 * (size_t)nb_rows(k_rows) + (size_t)KS * cout);
}

int tg_spectral_norm_fwd(const float* w, const float* u, float* w_bar, float* u_new, float* v, float* stats, int k_rows,
                         int cout, void* ws, size_t ws_bytes, void* stream) {
  TG_CHECK(w && u && w_bar && u_new && v && stats && k_rows > 0 && cout > 0 && cout <= 1024, TG_EINVAL,
           "tg_spectral_norm_fwd: bad arguments (K %d, cout %d)", k_rows, cout);
  TG_CHECK(ws && ws_bytes >= tg_spectral_norm_workspace(k_rows, cout), TG_EINVAL, "tg_spectral_norm_fwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int nb = nb_rows(k_rows);
  float* v_raw = (float*)ws;
  float* part_ss = v_raw + k_rows;
  float* upart = part_ss + 3 * nb;
  hipLaunchKernelGGL((sn_rowdot<false>), dim3(nb), dim3(256), 0, s, w, u, v_raw, part_ss, nullptr, nullptr, nullptr, nullptr,
                     k_rows, cout);
  hipLaunchKernelGGL(sn_coldot, dim3((cout + 63) / 64, KS), dim3(256), 0, s, w, v_raw, part_ss, nb, upart, k_rows, cout);
  const size_t total = (size_t)k_rows * cout;
  hipLaunchKernelGGL(sn_finish, dim3((unsigned)((total + 1023) / 1024)), dim3(256), 0, s, w, upart, v_raw, part_ss, nb, w_bar,
                     u_new, v, stats, k_rows, cout);
  TG_LAUNCH_CHECK("tg_spectral_norm_fwd");
  return TG_OK;
}

int tg_spectral_norm_bwd(const float* g_wbar, const float* w, const float* u, const float* u_new, const float* v,
                         const float* stats, float* gw, int accumulate, int k_rows, int cout, void* ws, size_t ws_bytes,
                         void* stream) {
  TG_CHECK(g_wbar && w && u && u_new && v && stats && gw && k_rows > 0 && cout > 0, TG_EINVAL,
           "tg_spectral_norm_bwd: bad arguments");
  TG_CHECK(ws && ws_bytes >= tg_spectral_norm_workspace(k_rows, cout), TG_EINVAL, "tg_spectral_norm_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int nb = nb_rows(k_rows);
  float* a = (float*)ws;
  float* part_s = a + k_rows + nb;
  float* part_va = part_s + nb;
  hipLaunchKernelGGL((sn_rowdot<true>), dim3(nb), dim3(256), 0, s, w, u_new, a, nullptr, g_wbar, v, part_s, part_va, k_rows,
                     cout);
  const size_t total = (size_t)k_rows * cout;
  hipLaunchKernelGGL(sn_bwd_apply, dim3((unsigned)((total + 1023) / 1024)), dim3(256), 0, s, g_wbar, u, u_new, v, a, part_s,
                     part_va, nb, stats, gw, accumulate, k_rows, cout);
  TG_LAUNCH_CHECK("tg_spectral_norm_bwd");
  return TG_OK;
}

// ---- many kernels, three launches (the trainer's prepare_run under --spectral_norm: 60 matrices in config 4) ----------
size_t tg_sn_table_bytes(int njobs) { return njobs > 0 ? (size_t)njobs * sizeof(SnJob) : 0; }

// Fills job j of a HOST table (copied to the device by the caller) and advances the three running block totals.
int tg_sn_table_fill(int j, const float* w, const float* u, float* w_bar, float* u_new, float* v, float* stats, void* ws,
                     size_t ws_bytes, int k_rows, int cout, void* host_table, int32_t* totals) {
  TG_CHECK(j >= 0 && w && u && w_bar && u_new && v && stats && ws && host_table && totals && k_rows > 0 && cout > 0 && cout <= 1024,
           TG_EINVAL, "tg_sn_table_fill: bad arguments (K %d, cout %d)", k_rows, cout);
  TG_CHECK(ws_bytes >= tg_spectral_norm_workspace(k_rows, cout), TG_EINVAL, "tg_sn_table_fill: workspace too small");
  SnJob& J = ((SnJob*)host_table)[j];
  J.w = w; J.u = u; J.w_bar = w_bar; J.u_new = u_new; J.v = v; J.stats = stats; J.ws = (float*)ws;
  J.k_rows = k_rows; J.cout = cout; J.nb = nb_rows(k_rows);
  J.row0 = totals[0]; J.col0 = totals[1]; J.fin0 = totals[2];
  totals[0] += J.nb;
  totals[1] += ((cout + 63) / 64) * KS;
  totals[2] += (int)(((size_t)k_rows * cout + 1023) / 1024);
  return TG_OK;
}

int tg_spectral_norm_fwd_multi(const void* table, int njobs, int row_blocks, int col_blocks, int fin_blocks, void* stream) {
  TG_CHECK(table && njobs > 0 && row_blocks > 0 && col_blocks > 0 && fin_blocks > 0, TG_EINVAL,
           "tg_spectral_norm_fwd_multi: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sn_rowdot_multi, dim3(row_blocks), dim3(256), 0, s, (const SnJob*)table, njobs);
  hipLaunchKernelGGL(sn_coldot_multi, dim3(col_blocks), dim3(256), 0, s, (const SnJob*)table, njobs);
  hipLaunchKernelGGL(sn_finish_multi, dim3(fin_blocks), dim3(256), 0, s, (const SnJob*)table, njobs);
  TG_LAUNCH_CHECK("tg_spectral_norm_fwd_multi");
  return TG_OK;
}

int tg_sn_assign_u(const void* table, int njobs, void* stream) {
  TG_CHECK(table && njobs > 0, TG_EINVAL, "tg_sn_assign_u: bad arguments");
  hipLaunchKernelGGL(sn_assign_u_multi, dim3(njobs), dim3(256), 0, (hipStream_t)stream, (const SnJob*)table);
  TG_LAUNCH_CHECK("tg_sn_assign_u");
  return TG_OK;
}

}  // extern "C"
