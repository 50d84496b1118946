// common.h -- shared device/host helpers for libflowdec_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "flowdec_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// ---- error plumbing (api.hip) -----------------------------------------------------------------
int fd_set_error(int code, const char* fmt, ...);

#define FD_HIP(expr)                                                                                 \
  do {                                                                                               \
    hipError_t _e = (expr);                                                                          \
    if (_e != hipSuccess)                                                                            \
      return fd_set_error(FD_ERUNTIME, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

#define FD_REQUIRE(cond, ...)                                    \
  do {                                                           \
    if (!(cond)) return fd_set_error(FD_EINVAL, __VA_ARGS__);    \
  } while (0)

#define FD_LAUNCH_CHECK() FD_HIP(hipGetLastError())

#define FD_TRY(expr)              \
  do {                            \
    int _rc = (expr);             \
    if (_rc != FD_OK) return _rc; \
  } while (0)

static inline hipStream_t fd_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline size_t fd_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
static inline int fd_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline size_t fd_dtype_size(int dtype) { return dtype == FD_BF16 ? 2 : 4; }

// ---- device helpers ---------------------------------------------------------------------------
// x * sigmoid(x) with the hardware reciprocal (1 ulp) instead of an IEEE division (10 VALU instructions)
__device__ __forceinline__ float fd_silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

template <typename T>
struct Elem;
template <>
struct Elem<float> {
  static constexpr int dtype = FD_F32;
  __device__ static float ld(const float* p) { return *p; }
  __device__ static void st(float* p, float v) { *p = v; }
};
template <>
struct Elem<bf16> {
  static constexpr int dtype = FD_BF16;
  __device__ static float ld(const bf16* p) { return (float)*p; }
  __device__ static void st(bf16* p, float v) { *p = (bf16)v; }
};

// Load / store N (= 4 or 8) consecutive elements as floats with the widest aligned access.
template <typename T, int N>
__device__ __forceinline__ void fd_load_vec(const T* p, float (&v)[N]);
template <>
__device__ __forceinline__ void fd_load_vec<float, 4>(const float* p, float (&v)[4]) {
  f32x4 t = *reinterpret_cast<const f32x4*>(p);
  v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
}
template <>
__device__ __forceinline__ void fd_load_vec<float, 8>(const float* p, float (&v)[8]) {
  f32x4 t0 = *reinterpret_cast<const f32x4*>(p), t1 = *reinterpret_cast<const f32x4*>(p + 4);
  v[0] = t0[0]; v[1] = t0[1]; v[2] = t0[2]; v[3] = t0[3];
  v[4] = t1[0]; v[5] = t1[1]; v[6] = t1[2]; v[7] = t1[3];
}
template <>
__device__ __forceinline__ void fd_load_vec<bf16, 4>(const bf16* p, float (&v)[4]) {
  bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
  v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
}
template <>
__device__ __forceinline__ void fd_load_vec<bf16, 8>(const bf16* p, float (&v)[8]) {
  bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
}
template <typename T, int N>
__device__ __forceinline__ void fd_store_vec(T* p, const float (&v)[N]);
template <>
__device__ __forceinline__ void fd_store_vec<float, 4>(float* p, const float (&v)[4]) {
  f32x4 t = {v[0], v[1], v[2], v[3]};
  *reinterpret_cast<f32x4*>(p) = t;
}
template <>
__device__ __forceinline__ void fd_store_vec<float, 8>(float* p, const float (&v)[8]) {
  f32x4 t0 = {v[0], v[1], v[2], v[3]}, t1 = {v[4], v[5], v[6], v[7]};
  *reinterpret_cast<f32x4*>(p) = t0;
  *reinterpret_cast<f32x4*>(p + 4) = t1;
}
template <>
__device__ __forceinline__ void fd_store_vec<bf16, 4>(bf16* p, const float (&v)[4]) {
  bf16x4 t = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
  *reinterpret_cast<bf16x4*>(p) = t;
}
template <>
__device__ __forceinline__ void fd_store_vec<bf16, 8>(bf16* p, const float (&v)[8]) {
  bf16x8 t;
#pragma unroll
  for (int i = 0; i < 8; ++i) t[i] = (bf16)v[i];
  *reinterpret_cast<bf16x8*>(p) = t;
}

// block-wide sum of a double over a 256-thread block (used by the statistics kernels)
__device__ __forceinline__ double fd_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ __forceinline__ float fd_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ __forceinline__ float fd_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o, 64));
  return v;
}

// ---- internal cross-file entry points -----------------------------------------------------------
struct fd_conv_timing;  // model.hip
