#include <hip/hip_runtime.h>
__global__ void k(unsigned* out) {
  unsigned a = threadIdx.x, b = threadIdx.x + 100;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[threadIdx.x] = r[0]; out[64 + threadIdx.x] = r[1];
}
int main(){ unsigned* d; (void)hipMalloc(&d, 512); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); unsigned h[128]; (void)hipMemcpy(h,d,512,hipMemcpyDeviceToHost); for(int i=0;i<128;i+=8) printf("%u %u %u %u %u %u %u %u\n",h[i],h[i+1],h[i+2],h[i+3],h[i+4],h[i+5],h[i+6],h[i+7]); }
