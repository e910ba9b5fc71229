// Direct (one thread per output element) stride-1 convolution kernels: forward, backward-data,
// backward-weight.  Any shape, fp32 or bf16 activations, fp32 accumulation.  This is the exact-f32
// parity path and the on-device cross-check for the MFMA kernels at sizes the CPU oracle cannot
// reach.  Replaces Conv2D / Conv2DBackpropInput / Conv2DBackpropFilter behind
// nets/pggan_utils.py:316-320 (reference, TF-1.8 kernels).
#include "tg_common.h"

namespace {

// y[n,oy,ox,co] = sum_{ky,kx,ci} x[n,oy+ky-pt,ox+kx-pl,ci] * w[ky,kx,ci,co]
template <typename T>
__global__ void conv_fwd_direct(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                T* __restrict__ y, TgConvDesc d) {
  const int64_t total = (int64_t)d.n * d.hout * d.wout * d.cout;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int co = (int)(i % d.cout);
    int64_t p = i / d.cout;
    const int ox = (int)(p % d.wout);
    p /= d.wout;
    const int oy = (int)(p % d.hout);
    const int n = (int)(p / d.hout);
    float acc = 0.f;
    for (int ky = 0; ky < d.kh; ++ky) {
      const int iy = oy + ky - d.pad_t;
      if (iy < 0 || iy >= d.hin) continue;
      for (int kx = 0; kx < d.kw; ++kx) {
        const int ix = ox + kx - d.pad_l;
        if (ix < 0 || ix >= d.win) continue;
        const T* xp = x + (((int64_t)n * d.hin + iy) * d.win + ix) * d.cin;
        const float* wp = w + ((int64_t)(ky * d.kw + kx) * d.cin) * d.cout + co;
        for (int ci = 0; ci < d.cin; ++ci) acc = fmaf(ld(xp + ci), rnd<T>(wp[(int64_t)ci * d.cout]), acc);
      }
    }
    if (d.epilogue & TG_EPI_BIAS) acc += bias[co];
    if (d.epilogue & TG_EPI_LRELU) acc = lrelu_f(acc, d.lrelu_alpha);
    st(y + i, acc);
  }
}

// gx[n,iy,ix,ci] = sum_{ky,kx,co} gy[n,iy+pt-ky,ix+pl-kx,co] * w[ky,kx,ci,co]
template <typename T>
__global__ void conv_bwd_data_direct(const T* __restrict__ gy, const float* __restrict__ w, T* __restrict__ gx,
                                     TgConvDesc d) {
  const int64_t total = (int64_t)d.n * d.hin * d.win * d.cin;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % d.cin);
    int64_t p = i / d.cin;
    const int ix = (int)(p % d.win);
    p /= d.win;
    const int iy = (int)(p % d.hin);
    const int n = (int)(p / d.hin);
    float acc = 0.f;
    for (int ky = 0; ky < d.kh; ++ky) {
      const int oy = iy + d.pad_t - ky;
      if (oy < 0 || oy >= d.hout) continue;
      for (int kx = 0; kx < d.kw; ++kx) {
        const int ox = ix + d.pad_l - kx;
        if (ox < 0 || ox >= d.wout) continue;
        const T* gp = gy + (((int64_t)n * d.hout + oy) * d.wout + ox) * d.cout;
        const float* wp = w + ((int64_t)(ky * d.kw + kx) * d.cin + ci) * d.cout;
        for (int co = 0; co < d.cout; ++co) acc = fmaf(ld(gp + co), rnd<T>(wp[co]), acc);
      }
    }
    st(gx + i, acc);
  }
}

// gw[ky,kx,ci,co] += sum over a chunk of output pixels of x[.., ci] * gy[.., co].  With `slab` every pixel chunk
// writes its partial result to slab[chunk][.] and the chunks are summed in order by the slab reduction (the exact-parity
// path: no float atomics, the result does not depend on block timing); without workspace the chunks are added atomically.
template <typename T>
__global__ void conv_bwd_weight_direct(const T* __restrict__ x, const T* __restrict__ gy, float* __restrict__ gw,
                                       float* __restrict__ slab, TgConvDesc d, int pix_per_chunk) {
  const int64_t nw = (int64_t)d.kh * d.kw * d.cin * d.cout;
  const int64_t npix = (int64_t)d.n * d.hout * d.wout;
  const int64_t p0 = (int64_t)blockIdx.y * pix_per_chunk;
  const int64_t p1 = (p0 + pix_per_chunk < npix) ? p0 + pix_per_chunk : npix;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nw; i += (int64_t)gridDim.x * blockDim.x) {
    const int co = (int)(i % d.cout);
    int64_t q = i / d.cout;
    const int ci = (int)(q % d.cin);
    q /= d.cin;
    const int kx = (int)(q % d.kw);
    const int ky = (int)(q / d.kw);
    float acc = 0.f;
    for (int64_t p = p0; p < p1; ++p) {
      const int ox = (int)(p % d.wout);
      const int64_t r = p / d.wout;
      const int oy = (int)(r % d.hout);
      const int n = (int)(r / d.hout);
      const int iy = oy + ky - d.pad_t, ix = ox + kx - d.pad_l;
      if (iy < 0 || iy >= d.hin || ix < 0 || ix >= d.win) continue;
      acc = fmaf(ld(x + (((int64_t)n * d.hin + iy) * d.win + ix) * d.cin + ci), ld(gy + p * d.cout + co), acc);
    }
    if (slab) slab[(size_t)blockIdx.y * nw + i] = acc;
    else atomicAdd(gw + i, acc);
  }
}

}  // namespace

int tg_conv2d_fwd_direct(const TgConvDesc* d, const void* x, const void* w, const float* bias, void* y,
                         hipStream_t s) {
  const int64_t total = (int64_t)d->n * d->hout * d->wout * d->cout;
  const int grid = tg_grid_for(total, 256, 256 * 64);
  tg_note_kernel("conv_fwd_direct");
  TG_DISPATCH_DTYPE(d->dtype, "tg_conv2d_fwd", {
    hipLaunchKernelGGL(conv_fwd_direct<T>, dim3(grid), dim3(256), 0, s, (const T*)x, (const float*)w, bias, (T*)y, *d);
  });
  TG_LAUNCH_CHECK("tg_conv2d_fwd(direct)");
  return TG_OK;
}

int tg_conv2d_bwd_data_direct(const TgConvDesc* d, const void* gy, const void* w, void* gx, hipStream_t s) {
  const int64_t total = (int64_t)d->n * d->hin * d->win * d->cin;
  const int grid = tg_grid_for(total, 256, 256 * 64);
  tg_note_kernel("conv_bwd_data_direct");
  TG_DISPATCH_DTYPE(d->dtype, "tg_conv2d_bwd_data", {
    hipLaunchKernelGGL(conv_bwd_data_direct<T>, dim3(grid), dim3(256), 0, s, (const T*)gy, (const float*)w, (T*)gx,
                       *d);
  });
  TG_LAUNCH_CHECK("tg_conv2d_bwd_data(direct)");
  return TG_OK;
}

int tg_wgrad_slab_reduce(const float* slab, float* gw, int64_t nw, int nslices, int accumulate, hipStream_t s);

static constexpr int DIRECT_PIX_PER_CHUNK = 2048;

size_t tg_conv2d_bwd_weight_workspace_direct(const TgConvDesc* d) {
  const int64_t nw = (int64_t)d->kh * d->kw * d->cin * d->cout;
  const int64_t npix = (int64_t)d->n * d->hout * d->wout;
  return (size_t)((npix + DIRECT_PIX_PER_CHUNK - 1) / DIRECT_PIX_PER_CHUNK) * nw * sizeof(float);
}

int tg_conv2d_bwd_weight_direct(const TgConvDesc* d, const void* x, const void* gy, float* gw, int accumulate,
                                hipStream_t s, void* ws, size_t ws_bytes) {
  const int64_t nw = (int64_t)d->kh * d->kw * d->cin * d->cout;
  const int64_t npix = (int64_t)d->n * d->hout * d->wout;
  float* slab = (ws && ws_bytes >= tg_conv2d_bwd_weight_workspace_direct(d)) ? (float*)ws : nullptr;
  if (!accumulate && !slab) {
    int rc = tg_zero_async(gw, nw * sizeof(float), nullptr, 0, s);
    if (rc) return rc;
  }
  tg_note_kernel("conv_bwd_weight_direct");
  const int pix_per_chunk = DIRECT_PIX_PER_CHUNK;
  const int gy_chunks = (int)((npix + pix_per_chunk - 1) / pix_per_chunk);
  const int gx = tg_grid_for(nw, 256, 4096);
  TG_CHECK(gy_chunks <= 65535, TG_EINVAL, "tg_conv2d_bwd_weight(direct): too many pixels (%lld)", (long long)npix);
  TG_DISPATCH_DTYPE(d->dtype, "tg_conv2d_bwd_weight", {
    hipLaunchKernelGGL(conv_bwd_weight_direct<T>, dim3(gx, gy_chunks), dim3(256), 0, s, (const T*)x, (const T*)gy, gw,
                       slab, *d, pix_per_chunk);
  });
  TG_LAUNCH_CHECK("tg_conv2d_bwd_weight(direct)");
  if (slab) return tg_wgrad_slab_reduce(slab, gw, nw, gy_chunks, accumulate, s);
  return TG_OK;
}
