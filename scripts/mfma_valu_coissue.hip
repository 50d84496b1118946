// Do vector-ALU instructions of ANOTHER wave (or of the same wave) run in the shadow of an MFMA on a gfx950 SIMD?  Asked for
// v_mfma_f32_16x16x4_f32 (float32: conv_wino44f.hip) and for v_mfma_f32_32x32x16_f16 (fp16: conv_wino4.hip).
// One workgroup of 512 threads per CU (two waves per SIMD).  Modes:
//   M   : every wave issues NM MFMAs per iteration (8 independent accumulators)
//   V   : every wave issues NV dependent-free v_fma_f32 per iteration
//   MV  : waves 0-3 issue MFMAs, waves 4-7 (the SIMD partners) issue v_fma        -- conv_wino44f.hip's schedule
//   MIX : every wave alternates 1 MFMA with NV / NM v_fma                         -- the same-wave interleave of conv_wino4.hip
// hipcc --offload-arch=gfx950 -O3 scripts/mfma_valu_coissue.hip -o scripts/bin/mfma_valu_coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
constexpr int NM = 64, NV = 256;

template <bool F16> struct Acc { typedef f32x4 T; };
template <> struct Acc<true> { typedef f32x16 T; };
template <bool F16>
__device__ __forceinline__ void mfma1(typename Acc<F16>::T& acc, float a, float b, f16x8 ha, f16x8 hb) {
  if constexpr (F16) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc, 0, 0, 0);
  else acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
}
template <bool F16>
__device__ __forceinline__ void mfmas(typename Acc<F16>::T (&acc)[4], float a, float b, f16x8 ha, f16x8 hb) {
#pragma unroll
  for (int i = 0; i < NM; ++i) mfma1<F16>(acc[i & 3], a, b, ha, hb);
}
__device__ __forceinline__ void valus(float (&v)[8], float a, float b) {
#pragma unroll
  for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[i & 7]) : "v"(a), "v"(b));
}
template <int MODE, bool F16>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  typename Acc<F16>::T acc[4];
  float v[8];
  for (int i = 0; i < 4; ++i) acc[i] = 0.f;
  for (int i = 0; i < 8; ++i) v[i] = 0.f;
  const unsigned h = threadIdx.x * 2654435761u;
  const float a = (float)(h & 15) * 0.1f, b = (float)((h >> 4) & 15) * 0.05f;
  f16x8 ha, hb;
  for (int e = 0; e < 8; ++e) { ha[e] = (_Float16)((float)((h >> e) & 7) * 0.25f); hb[e] = (_Float16)((float)((h >> (e + 3)) & 7) * 0.125f); }
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) mfmas<F16>(acc, a, b, ha, hb);
    else if (MODE == 1) valus(v, a, b);
    else if (MODE == 2) { if (wave < 4) mfmas<F16>(acc, a, b, ha, hb); else valus(v, a, b); }
    else {
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        mfma1<F16>(acc[i & 3], a, b, ha, hb);
#pragma unroll
        for (int j = 0; j < NV / NM; ++j) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[(i * (NV / NM) + j) & 7]) : "v"(a), "v"(b));
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += v[i];
  if (s == 123.456f) out[threadIdx.x] = s;
}
template <int MODE, bool F16>
float run(float* o, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, F16>), dim3(256), dim3(512), 0, 0, o, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  return ms;
}
template <bool F16>
void report(const char* name, float* o, int iters) {
  const float m = run<0, F16>(o, iters), v = run<1, F16>(o, iters), mv = run<2, F16>(o, iters), mix = run<3, F16>(o, iters);
  printf("%s, per iteration and SIMD (two waves):\n", name);
  printf("  M   = 2 x %d MFMA                               %7.3f ms  (%.1f ns per MFMA)\n", NM, m, m * 1e6 / (2.0 * NM * iters));
  printf("  V   = 2 x %d v_fma                             %7.3f ms  (%.2f ns per v_fma)\n", NV, v, v * 1e6 / (2.0 * NV * iters));
  printf("  MV  = one wave %d MFMA | its partner %d v_fma   %7.3f ms  (halves one after the other %.3f, perfect overlap %.3f)\n", NM, NV, mv, 0.5f * (m + v), 0.5f * (m > v ? m : v));
  printf("  MIX = 2 x (%d MFMA + %d v_fma, 1 : %d interleaved) %7.3f ms  (one after the other %.3f, perfect overlap %.3f)\n", NM, NV, NV / NM, mix, m + v, m > v ? m : v);
}
int main() {
  float* o; (void)hipMalloc(&o, 4096);
  report<false>("v_mfma_f32_16x16x4_f32", o, 4000);
  report<true>("v_mfma_f32_32x32x16_f16", o, 4000);
  return 0;
}
