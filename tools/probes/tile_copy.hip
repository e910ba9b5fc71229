// Probe: does the 2-D tile access pattern of the thin-layer conv kernels cost HBM bandwidth?  Same bytes three ways on a
// [n, 256, 256, 16] bf16 tensor (32 B per pixel): (a) flat streaming copy, (b) per-workgroup walk over 8x16-pixel tiles
// reading the 10x18 halo and writing the 8x16 tile (what conv_tile_wres does, minus the MFMAs), (c) per-workgroup walk
// down full-width row strips (2 rows per step, rolling window: every input row read once).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define HC(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__global__ __launch_bounds__(256) void flat_copy(const u32x4* __restrict__ x, u32x4* __restrict__ y, size_t nvec) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) y[i] = x[i];
}

// tiles of 8 rows x 16 cols; halo 10 x 18 pixels x 2 vectors (32 B) = 360 vectors; output 128 px x 2 = 256 vectors
__global__ __launch_bounds__(256) void tile_copy(const u32x4* __restrict__ x, u32x4* __restrict__ y, int n, int tiles_per_wg) {
  __shared__ u32x4 s[360];
  const int H = 256, W = 256, tx_n = 16, ty_n = 32, total = n * tx_n * ty_n;
  const int t0 = blockIdx.x * tiles_per_wg;
  for (int t = t0; t < t0 + tiles_per_wg && t < total; ++t) {
    const int tx = t % tx_n, ty = (t / tx_n) % ty_n, img = t / (tx_n * ty_n);
    u32x4 r[2];
    for (int k = 0; k < 2; ++k) {
      const int v = threadIdx.x + k * 256;
      u32x4 z = {0, 0, 0, 0};
      if (v < 360) {
        const int px = v >> 1, part = v & 1, hy = px / 18 - 1 + ty * 8, hx = px % 18 - 1 + tx * 16;
        if (hy >= 0 && hy < H && hx >= 0 && hx < W) z = x[(((size_t)img * H + hy) * W + hx) * 2 + part];
      }
      r[k] = z;
    }
    __syncthreads();
    s[threadIdx.x] = r[0];
    if (threadIdx.x + 256 < 360) s[threadIdx.x + 256] = r[1];
    __syncthreads();
    const int px = threadIdx.x >> 1, part = threadIdx.x & 1, oy = px >> 4, ox = px & 15;
    u32x4 a = s[((oy + 1) * 18 + ox + 1) * 2 + part], b = s[(oy * 18 + ox) * 2 + part];
    a[0] ^= b[1];
    y[(((size_t)img * H + ty * 8 + oy) * W + tx * 16 + ox) * 2 + part] = a;
  }
}

// strips: a workgroup owns `rows_per_wg` consecutive rows of one image, walks them 1 row at a time (512 vectors = 8 KB)
__global__ __launch_bounds__(256) void row_copy(const u32x4* __restrict__ x, u32x4* __restrict__ y, int rows_per_wg) {
  const size_t row0 = (size_t)blockIdx.x * rows_per_wg;
  u32x4 r0[2], r1[2];
  for (int k = 0; k < 2; ++k) r0[k] = x[row0 * 512 + threadIdx.x + k * 256];
  for (int rr = 0; rr < rows_per_wg; ++rr) {
    if (rr + 1 < rows_per_wg)
      for (int k = 0; k < 2; ++k) r1[k] = x[(row0 + rr + 1) * 512 + threadIdx.x + k * 256];
    for (int k = 0; k < 2; ++k) y[(row0 + rr) * 512 + threadIdx.x + k * 256] = r0[k];
    for (int k = 0; k < 2; ++k) r0[k] = r1[k];
  }
}

int main() {
  const int n = 64;
  const size_t nvec = (size_t)n * 256 * 256 * 2, bytes = nvec * 16;
  u32x4 *x, *y;
  HC(hipMalloc(&x, bytes)); HC(hipMalloc(&y, bytes));
  HC(hipMemset(x, 1, bytes));
  hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto fn) {
    fn(); fn(); HC(hipDeviceSynchronize());
    HC(hipEventRecord(e0)); for (int i = 0; i < 10; ++i) fn(); HC(hipEventRecord(e1)); HC(hipEventSynchronize(e1));
    float ms; HC(hipEventElapsedTime(&ms, e0, e1));
    printf("%-28s %8.1f us  %7.0f GB/s (read + write)\n", name, 1e3 * ms / 10, 2.0 * bytes / (ms / 10 * 1e-3) * 1e-9);
  };
  timeit("flat copy", [&] { hipLaunchKernelGGL(flat_copy, dim3(4096), dim3(256), 0, 0, x, y, nvec); });
  for (int tpw : {1, 4, 16, 32})
    timeit(tpw == 1 ? "tiles 8x16, 1 per wg" : tpw == 4 ? "tiles 8x16, 4 per wg" : tpw == 16 ? "tiles 8x16, 16 per wg" : "tiles 8x16, 32 per wg",
           [&] { hipLaunchKernelGGL(tile_copy, dim3((n * 512 + tpw - 1) / tpw), dim3(256), 0, 0, x, y, n, tpw); });
  for (int rpw : {4, 16, 64})
    timeit(rpw == 4 ? "rows, 4 per wg" : rpw == 16 ? "rows, 16 per wg" : "rows, 64 per wg",
           [&] { hipLaunchKernelGGL(row_copy, dim3(n * 256 / rpw), dim3(256), 0, 0, x, y, rpw); });
  return 0;
}
