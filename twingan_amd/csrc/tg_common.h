// Shared device/host helpers for libtwingan_hip.so (gfx950 only).
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/twingan_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define TG_WAVE 64

// ---- error plumbing -------------------------------------------------------------------------
void tg_set_error(const char* fmt, ...);

#define TG_CHECK(cond, code, ...)  \
  do {                             \
    if (!(cond)) {                 \
      tg_set_error(__VA_ARGS__);   \
      return (code);               \
    }                              \
  } while (0)

#define TG_LAUNCH_CHECK(name)                                                     \
  do {                                                                            \
    hipError_t e__ = hipGetLastError();                                           \
    if (e__ != hipSuccess) {                                                      \
      tg_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));        \
      return TG_ELAUNCH;                                                          \
    }                                                                             \
  } while (0)

// Records the name of the kernel variant a dispatch picked (thread-local; read back with tg_last_kernel()), so
// that host-side per-launch timing can be attributed to the same kernel symbols rocprofv3 reports.
void tg_note_kernel(const char* fmt, ...);

// Zero-fills up to two fp32-aligned device buffers with ONE kernel launch on `s`.  Used instead of
// hipMemsetAsync everywhere: memset nodes captured into a hipGraph were observed to run out of order with
// the kernels that accumulate into the buffer (ROCm 7.2), corrupting replayed steps.
int tg_zero_async(void* a, size_t a_bytes, void* b, size_t b_bytes, hipStream_t s);

static inline bool tg_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// "Done once per DEVICE" flags for per-device driver state (hipFuncSetAttribute applies to the current device only): `mask`
// is a static of the call site; returns true when the current device's bit was still clear and sets it.  Devices >= 64 are
// always "not done" (the attribute call is idempotent).  Launch paths are entered by one host thread per device.
static inline bool tg_first_on_device(unsigned long long* mask) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
  if (*mask & (1ull << dev)) return false;
  *mask |= 1ull << dev;
  return true;
}

// ---- scalar load/store by storage type ------------------------------------------------------
template <typename T> __device__ __forceinline__ float ld(const T* p);
template <> __device__ __forceinline__ float ld<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld<bf16>(const bf16* p) { return (float)*p; }
template <> __device__ __forceinline__ float ld<f16>(const f16* p) { return (float)*p; }
template <typename T> __device__ __forceinline__ void st(T* p, float v);
template <> __device__ __forceinline__ void st<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st<bf16>(bf16* p, float v) { *p = (bf16)v; }
template <> __device__ __forceinline__ void st<f16>(f16* p, float v) { *p = (f16)v; }

// round-trip through the storage type (so fp32 math sees what a bf16 store would keep)
template <typename T> __device__ __forceinline__ float rnd(float v);
template <> __device__ __forceinline__ float rnd<float>(float v) { return v; }
template <> __device__ __forceinline__ float rnd<bf16>(float v) { return (float)(bf16)v; }
template <> __device__ __forceinline__ float rnd<f16>(float v) { return (float)(f16)v; }

// ---- 16-byte vectors of the storage type ----------------------------------------------------
template <typename T> struct Vec16;   // 16 bytes worth of T
template <> struct Vec16<float> {
  static constexpr int N = 4;
  f32x4 v;
  __device__ __forceinline__ float get(int i) const { return v[i]; }
  __device__ __forceinline__ void set(int i, float x) { v[i] = x; }
};
template <> struct Vec16<bf16> {
  static constexpr int N = 8;
  bf16x8 v;
  __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
  __device__ __forceinline__ void set(int i, float x) { v[i] = (bf16)x; }
};
template <> struct Vec16<f16> {
  static constexpr int N = 8;
  f16x8 v;
  __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
  __device__ __forceinline__ void set(int i, float x) { v[i] = (f16)x; }
};
template <typename T> __device__ __forceinline__ Vec16<T> ldv(const T* p) {
  return *reinterpret_cast<const Vec16<T>*>(p);
}
template <typename T> __device__ __forceinline__ void stv(T* p, const Vec16<T>& v) {
  *reinterpret_cast<Vec16<T>*>(p) = v;
}

// Cache policy of the big streaming stores (the `aux` operand of a raw buffer store: 0 plain, 2 nt, 16 sc1, 17 sc0 sc1).
// A compile-time switch for A/B builds (make CXXFLAGS=... -DTG_STORE_AUX=16): see DESIGN.md section 8c for what it measured.
#ifndef TG_STORE_AUX
#define TG_STORE_AUX 0
#endif
// 16 bytes to base[elem_off ...] with that policy; `base` must be wave-uniform, the byte offset below 4 GiB
template <typename T> __device__ __forceinline__ void stv_stream(T* base, unsigned elem_off, const Vec16<T>& v) {
#if TG_STORE_AUX
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0xffffffffu, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(bf16x8, v.v), r, elem_off * (unsigned)sizeof(T), 0, TG_STORE_AUX);
#else
  stv(base + elem_off, v);
#endif
}

// Which 16-bit format the MFMA kernels of the current C-ABI call see (thread-local; set by the entry point from the
// descriptor's / the call's dtype, read by the launchers that pick the template variant): false = bfloat16, true = half.
bool tg_elem_f16();
void tg_set_elem_f16(bool f16);

// ---- the two 16-bit storage formats of the MFMA kernels ---------------------------------------
// The kernels move activations and weight packs as opaque 16-byte vectors of eight 16-bit lanes (typed bf16x8 for
// historical reasons); the element format only matters where a value meets arithmetic: the MFMA instruction, the
// float -> 16-bit packing of an epilogue, the 16-bit -> float unpacking of the statistics / pool epilogues.
// F16 = false: bfloat16; true: IEEE half (TG_F16).
template <bool F16>
__device__ __forceinline__ f32x16 mfma_32x32x16(bf16x8 a, bf16x8 b, f32x16 c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <bool F16>
__device__ __forceinline__ f32x4 mfma_16x16x32(bf16x8 a, bf16x8 b, f32x4 c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <bool F16>
__device__ __forceinline__ unsigned pack16x2(float lo, float hi) {
  if constexpr (F16) {
    f16x2 v;
    v[0] = (f16)lo;
    v[1] = (f16)hi;
    return __builtin_bit_cast(unsigned, v);
  } else {
    bf16x2 v;
    v[0] = (bf16)lo;
    v[1] = (bf16)hi;
    return __builtin_bit_cast(unsigned, v);
  }
}
template <bool F16>
__device__ __forceinline__ float unpack16_lo(unsigned p) {
  if constexpr (F16) return (float)__builtin_bit_cast(f16x2, p)[0];
  else return __builtin_bit_cast(float, p << 16);
}
template <bool F16>
__device__ __forceinline__ float unpack16_hi(unsigned p) {
  if constexpr (F16) return (float)__builtin_bit_cast(f16x2, p)[1];
  else return __builtin_bit_cast(float, p & 0xffff0000u);
}
// 1.0 in either format, twice (the all-ones MFMA operand of the bias-gradient trick)
template <bool F16> constexpr unsigned ones16x2() { return F16 ? 0x3c003c00u : 0x3f803f80u; }

// ---- reductions -----------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-wide sum; result valid in every thread.  `red` = __shared__ float[>= blockDim/64].
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += red[i];
  return r;
}

__device__ __forceinline__ float lrelu_f(float x, float a) { return fmaxf(a * x, x); }

// The fp32 storage type is the exact-parity path: every sum that ends in one number per channel / sample / tensor is
// taken by ONE workgroup there (launchers pass a grid of 1), so no result depends on the order in which workgroups
// reach a float atomic -- the path is bit-reproducible.  The bf16 path keeps the wide grids and their fp32 atomics.
template <typename T> constexpr bool exact_path() { return sizeof(T) == 4; }

// TG_DETERMINISTIC=1 in the environment (or tg_set_deterministic(1)) gives the 16-bit storage types the same launch
// shapes: every launcher that asks exact_grid<T>() instead of exact_path<T>() then sums in a fixed order whatever the
// storage type, so two runs of a bf16 / fp16 step -- eagerly launched or replayed from a hipGraph -- are bit-identical.
// (exact_path<T>() stays the compile-time question "is this the fp32 parity path", e.g. for the run-time-flag variants
// of the normalisation kernels whose contraction pattern must not change.)
int tg_deterministic_mode();
template <typename T> inline bool exact_grid() { return sizeof(T) == 4 || tg_deterministic_mode() != 0; }

// The one kernel A/B left in the tree: TG_THIN16=0 sends the <= 16-output-channel 3x3 layers back to the 32-wide-block
// kernels (tests/test_gpu_ops.py compares the two families through it).  Read at every call.
static inline int tg_tune(const char* name, int dflt) {
  const char* v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}

static inline int tg_grid_for(int64_t work, int block, int max_blocks = 256 * 16) {
  int64_t g = (work + block - 1) / block;
  if (g < 1) g = 1;
  if (g > max_blocks) g = max_blocks;
  return (int)g;
}

// dtype dispatch helper
#define TG_DISPATCH_DTYPE(dtype, NAME, ...)                 \
  do {                                                      \
    if ((dtype) == TG_F32) {                                \
      using T = float;                                      \
      __VA_ARGS__                                           \
    } else if ((dtype) == TG_BF16) {                        \
      using T = bf16;                                       \
      __VA_ARGS__                                           \
    } else if ((dtype) == TG_F16) {                         \
      using T = f16;                                        \
      __VA_ARGS__                                           \
    } else {                                                \
      tg_set_error("%s: unsupported dtype %d", NAME, dtype); \
      return TG_EINVAL;                                     \
    }                                                       \
  } while (0)
