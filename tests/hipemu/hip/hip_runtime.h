// hipemu: a stand-in for <hip/hip_runtime.h> that lets twingan_amd/csrc/*.hip be compiled as plain host C++ and RUN on
// CPU cores, so that the kernels' index arithmetic, LDS staging, lane exchanges and MFMA operand layouts can be checked
// against the oracle without a GPU.  TEST INFRASTRUCTURE ONLY: nothing under twingan_amd/ includes or links this; the
// product library is built by hipcc for gfx950 (twingan_amd/csrc/Makefile) and has no CPU path.
//
// Execution model (emu.cpp): a launch runs its workgroups one after the other; the threads of a workgroup are fibers
// scheduled round-robin on one OS thread.  __syncthreads() and every wave-level operation (DPP, permlane swaps, shuffles, LDS
// transpose reads, MFMA) are rendezvous points of the workgroup / of the lanes of a wave that issue the same instruction: a
// lane deposits its operands, waits for the others, and reads what the instruction would have delivered to it; lanes of a
// wave that diverged (a loop's last trips, a guarded reduction) are released group by group, as under partial EXEC masks.
// The lane layouts below are the gfx950 ones the kernels were written (and verified on hardware) against:
//   v_mfma_f32_32x32x16: A lane l = row l%32, k 8*(l/32)..+8;  B lane l = column l%32, same k;  D register r of lane l =
//                        row 8*(r/4) + 4*(l/32) + r%4, column l%32
//   v_mfma_f32_16x16x32: A lane l = row l%16, k 8*(l/16)..+8;  B likewise;  D register r = row 4*(l/16) + r, column l%16
//   v_permlane32_swap(vdst, src): vdst of lanes 32..63 <-> src of lanes 0..31 (returns {vdst, src});  permlane16_swap: odd
//                        16-lane rows of vdst <-> even rows of src
//   ds_read_b64_tr_b16:  inside a 16-lane group, lane i element j = element i%4 of what lane 4*j + i/4 addressed
//   DPP quad_perm:       lane i reads lane (i & ~3) | ((ctrl >> 2*(i&3)) & 3)
//   raw buffer access:   a dword whose byte offset + 4 exceeds num_records reads 0 / is not stored
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
static inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

typedef void* hipStream_t;
typedef void* hipEvent_t;
enum hipError_t { hipSuccess = 0, hipErrorUnknown = 999 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
enum { hipEventDisableTiming = 2 };
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "hipemu"; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
template <typename F> static inline hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) { return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }

namespace hipemu {

struct Idx { unsigned x, y, z; };
struct Fiber;
extern Fiber* cur;
extern Idx block_idx, block_dim, grid_dim;
extern unsigned char dyn_lds[];      // the dynamic shared memory of the running workgroup (160 KB)
const Idx& thread_idx();
int lane_id();
bool deposited(int lane);      // did `lane` take part in the wave operation this lane just completed (exited / absent lanes: no)

void launch(dim3 grid, dim3 block, size_t dyn_bytes, const std::function<void()>& body);
void sync_block();
// deposits `bytes` (<= 64) for this lane and waits for the lanes that issue the same instruction (`site`)
void wave_exchange(const void* mine, int bytes);
const unsigned char* peer(int lane);      // what `lane` deposited in the operation this lane just completed
// the instruction's identity is wave_exchange's return address: every wave primitive below is force-inlined into its caller
#define HIPEMU_PRIM inline __attribute__((always_inline))

template <typename T> inline T from_lane(int lane, int byte_off = 0) {
  T v;
  memcpy(&v, peer(lane) + byte_off, sizeof(T));
  return v;
}

struct Rsrc {
  unsigned char* base;
  uint32_t bytes;
};

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;

inline Rsrc make_buffer_rsrc(void* p, short, int num, int) { return Rsrc{(unsigned char*)p, (uint32_t)num}; }
inline uint32_t buf_dword(const Rsrc& r, uint64_t off) {
  uint32_t v = 0;
  if (off + 4 <= r.bytes) memcpy(&v, r.base + off, 4);
  return v;
}
inline u32x4_t buffer_load_b128(const Rsrc& r, unsigned voff, unsigned soff, int) {
  u32x4_t v;
  for (int i = 0; i < 4; ++i) v[i] = buf_dword(r, (uint64_t)voff + soff + 4 * i);
  return v;
}
inline u32x2_t buffer_load_b64(const Rsrc& r, unsigned voff, unsigned soff, int) {
  u32x2_t v;
  for (int i = 0; i < 2; ++i) v[i] = buf_dword(r, (uint64_t)voff + soff + 4 * i);
  return v;
}
inline unsigned char buffer_load_b8(const Rsrc& r, unsigned voff, unsigned soff, int) {
  const uint64_t off = (uint64_t)voff + soff;
  return off + 1 <= r.bytes ? r.base[off] : (unsigned char)0;
}
template <typename V> inline void buffer_store_b128(V val, const Rsrc& r, unsigned voff, unsigned soff, int) {
  static_assert(sizeof(V) == 16, "b128 store");
  unsigned char b[16];
  memcpy(b, &val, 16);
  for (int i = 0; i < 4; ++i) {
    const uint64_t off = (uint64_t)voff + soff + 4 * i;
    if (off + 4 <= r.bytes) memcpy(r.base + off, b + 4 * i, 4);
  }
}
template <typename V> inline void buffer_store_b64(V val, const Rsrc& r, unsigned voff, unsigned soff, int) {
  static_assert(sizeof(V) == 8, "b64 store");
  unsigned char b[8];
  memcpy(b, &val, 8);
  for (int i = 0; i < 2; ++i) {
    const uint64_t off = (uint64_t)voff + soff + 4 * i;
    if (off + 4 <= r.bytes) memcpy(r.base + off, b + 4 * i, 4);
  }
}
inline void buffer_store_b16(short val, const Rsrc& r, unsigned voff, unsigned soff, int) {
  const uint64_t off = (uint64_t)voff + soff;
  if (off + 2 <= r.bytes) memcpy(r.base + off, &val, 2);
}
inline void buffer_store_b8(char val, const Rsrc& r, unsigned voff, unsigned soff, int) {
  const uint64_t off = (uint64_t)voff + soff;
  if (off + 1 <= r.bytes) r.base[off] = (unsigned char)val;
}

HIPEMU_PRIM int update_dpp(int, int src, int ctrl, int, int, bool) {
  if (ctrl < 0 || ctrl > 0xff) {
    fprintf(stderr, "hipemu: DPP control 0x%x is not a quad_perm\n", ctrl);
    abort();
  }
  const int l = lane_id();
  wave_exchange(&src, 4);
  return from_lane<int>((l & ~3) | ((ctrl >> (2 * (l & 3))) & 3));
}

struct SwapResult {
  unsigned v[2];
  unsigned operator[](int i) const { return v[i]; }
};
template <int ROW> HIPEMU_PRIM SwapResult permlane_swap(unsigned vdst, unsigned src) {
  const int l = lane_id();
  const unsigned mine[2] = {vdst, src};
  wave_exchange(mine, 8);
  SwapResult r;
  if ((l / ROW) & 1) {      // odd row: vdst <- src of the even row below; src stays
    r.v[0] = from_lane<unsigned>(l - ROW, 4);
    r.v[1] = src;
  } else {                  // even row: vdst stays; src <- vdst of the odd row above
    r.v[0] = vdst;
    r.v[1] = from_lane<unsigned>(l + ROW, 0);
  }
  return r;
}

template <typename T> HIPEMU_PRIM T shfl_xor(T v, int mask) {
  static_assert(sizeof(T) <= 8, "shuffle operand");
  const int l = lane_id();
  wave_exchange(&v, (int)sizeof(T));
  return from_lane<T>((l ^ mask) & 63);
}
HIPEMU_PRIM int any_lane(int pred) {
  const int p = pred != 0;
  wave_exchange(&p, 4);
  int r = 0;
  for (int i = 0; i < 64; ++i)
    if (deposited(i)) r |= from_lane<int>(i);
  return r;
}

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
HIPEMU_PRIM s16x4_t ds_read_tr16_b64(const void* p) {
  const int l = lane_id(), g = l & ~15, i = l & 15;
  if ((uintptr_t)p & 7) {
    fprintf(stderr, "hipemu: ds_read_b64_tr_b16 address not 8-byte aligned (returns the aligned address's data on gfx950)\n");
    abort();
  }
  short mine[4];
  memcpy(mine, p, 8);
  wave_exchange(mine, 8);
  s16x4_t r;
  for (int j = 0; j < 4; ++j) r[j] = from_lane<short>(g + 4 * j + i / 4, 2 * (i % 4));
  return r;
}

typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
// E = the 16-bit element type of the operands (8 per lane)
template <typename E, typename V> HIPEMU_PRIM f32x16_t mfma_32x32x16(V a, V b, f32x16_t c) {
  static_assert(sizeof(V) == 16, "operand");
  const int l = lane_id();
  unsigned char mine[32];
  memcpy(mine, &a, 16);
  memcpy(mine + 16, &b, 16);
  wave_exchange(mine, 32);
  const int col = l & 31;
  float bk[16];
  for (int k = 0; k < 16; ++k) bk[k] = (float)from_lane<E>(col + 32 * (k / 8), 16 + 2 * (k % 8));
  for (int r = 0; r < 16; ++r) {
    const int row = 8 * (r / 4) + 4 * (l / 32) + r % 4;
    float s = c[r];
    for (int k = 0; k < 16; ++k) s = fmaf((float)from_lane<E>(row + 32 * (k / 8), 2 * (k % 8)), bk[k], s);
    c[r] = s;
  }
  return c;
}
template <typename E, typename V> HIPEMU_PRIM f32x4_t mfma_16x16x32(V a, V b, f32x4_t c) {
  static_assert(sizeof(V) == 16, "operand");
  const int l = lane_id();
  unsigned char mine[32];
  memcpy(mine, &a, 16);
  memcpy(mine + 16, &b, 16);
  wave_exchange(mine, 32);
  const int col = l & 15;
  float bk[32];
  for (int k = 0; k < 32; ++k) bk[k] = (float)from_lane<E>(col + 16 * (k / 8), 16 + 2 * (k % 8));
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * (l / 16) + r;
    float s = c[r];
    for (int k = 0; k < 32; ++k) s = fmaf((float)from_lane<E>(row + 16 * (k / 8), 2 * (k % 8)), bk[k], s);
    c[r] = s;
  }
  return c;
}

}  // namespace hipemu

#define threadIdx (hipemu::thread_idx())
#define blockIdx (hipemu::block_idx)
#define blockDim (hipemu::block_dim)
#define gridDim (hipemu::grid_dim)
#define warpSize 64

#define hipLaunchKernelGGL(kernel, grid, block, smem, stream, ...) \
  hipemu::launch((grid), (block), (size_t)(smem), [&]() { kernel(__VA_ARGS__); })

static inline void __syncthreads() { hipemu::sync_block(); }
template <typename T> static HIPEMU_PRIM T __shfl_xor(T v, int mask, int = 64) { return hipemu::shfl_xor(v, mask); }
static HIPEMU_PRIM int __any(int p) { return hipemu::any_lane(p); }
static inline float atomicAdd(float* p, float v) { const float o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { const int o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { const unsigned o = *p; *p = o + v; return o; }
static inline void __threadfence() {}
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
#define __expf expf
#define __logf logf
#include <algorithm>
using std::max;
using std::min;
static inline float __fdividef(float a, float b) { return a / b; }

typedef hipemu::Rsrc __amdgpu_buffer_rsrc_t;
#define __builtin_amdgcn_make_buffer_rsrc hipemu::make_buffer_rsrc
#define __builtin_amdgcn_raw_buffer_load_b128 hipemu::buffer_load_b128
#define __builtin_amdgcn_raw_buffer_load_b64 hipemu::buffer_load_b64
#define __builtin_amdgcn_raw_buffer_load_b8 hipemu::buffer_load_b8
#define __builtin_amdgcn_raw_buffer_store_b128 hipemu::buffer_store_b128
#define __builtin_amdgcn_raw_buffer_store_b64 hipemu::buffer_store_b64
#define __builtin_amdgcn_raw_buffer_store_b16 hipemu::buffer_store_b16
#define __builtin_amdgcn_raw_buffer_store_b8 hipemu::buffer_store_b8
#define __builtin_amdgcn_update_dpp hipemu::update_dpp
#define __builtin_amdgcn_permlane32_swap(a, b, fi, bc) hipemu::permlane_swap<32>((a), (b))
#define __builtin_amdgcn_permlane16_swap(a, b, fi, bc) hipemu::permlane_swap<16>((a), (b))
#define __builtin_amdgcn_ds_read_tr16_b64_v4i16(p) hipemu::ds_read_tr16_b64((const void*)(p))
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(a, b, c) ((void)0)
#define __builtin_amdgcn_exp2f exp2f
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) hipemu::mfma_32x32x16<__bf16>((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) hipemu::mfma_32x32x16<_Float16>((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z) hipemu::mfma_16x16x32<__bf16>((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, x, y, z) hipemu::mfma_16x16x32<_Float16>((a), (b), (c))
// LDS pointers are ordinary pointers here
#define address_space(n) aligned(2)
