#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void probe(int* out, int mode) {
  __shared__ __attribute__((aligned(16))) short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x;
  int off;  // element offset supplied by this lane
  if (mode == 0) off = 0;                       // uniform
  else if (mode == 1) off = l * 4;              // natural contiguous 8B per lane
  else if (mode == 2) off = (l & 15) * 64 + (l >> 4) * 4;   // lane->row(l&15) of a [16][64] matrix, 4-col group l>>4
  else off = (l & 15) * 16 + (l >> 4) * 4;      // [16 rows][16 cols]
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + off));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  int* d; hipMalloc(&d, 64 * 4 * 4);
  int h[256];
  for (int mode = 0; mode < 4; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) printf("  lane %2d: %5d %5d %5d %5d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  }
  return 0;
}
