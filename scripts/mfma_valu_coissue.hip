// Do vector-ALU instructions of ANOTHER wave (or of the same wave) run in the shadow of v_mfma_f32_16x16x4_f32 on a gfx950 SIMD?
// One workgroup of 512 threads per CU (two waves per SIMD).  Modes:
//   M   : every wave issues NM MFMAs per iteration (8 independent accumulators)
//   V   : every wave issues NV dependent-free v_fma_f32 per iteration
//   MV  : waves 0-3 issue MFMAs, waves 4-7 (the SIMD partners) issue v_fma        -- conv_wino44f.hip's schedule
//   MIX : every wave alternates 1 MFMA with NV / NM v_fma                         -- the same-wave interleave of conv_wino4.hip
// hipcc --offload-arch=gfx950 -O3 scripts/mfma_valu_coissue.hip -o scripts/bin/mfma_valu_coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int NM = 64, NV = 256;

__device__ __forceinline__ void mfmas(f32x4 (&acc)[8], float a, float b) {
#pragma unroll
  for (int i = 0; i < NM; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i & 7], 0, 0, 0);
}
__device__ __forceinline__ void valus(float (&v)[8], float a, float b) {
#pragma unroll
  for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[i & 7]) : "v"(a), "v"(b));
}
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  f32x4 acc[8];
  float v[8];
  for (int i = 0; i < 8; ++i) { acc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; v[i] = 0.f; }
  const unsigned h = threadIdx.x * 2654435761u;
  const float a = (float)(h & 15) * 0.1f, b = (float)((h >> 4) & 15) * 0.05f;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) mfmas(acc, a, b);
    else if (MODE == 1) valus(v, a, b);
    else if (MODE == 2) { if (wave < 4) mfmas(acc, a, b); else valus(v, a, b); }
    else {
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i & 7], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NV / NM; ++j) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[(i * (NV / NM) + j) & 7]) : "v"(a), "v"(b));
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i];
  if (s == 123.456f) out[threadIdx.x] = s;
}
template <int MODE>
float run(float* o, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 0, 0, o, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  return ms;
}
int main() {
  float* o; hipMalloc(&o, 4096);
  const int iters = 4000;
  const float m = run<0>(o, iters), v = run<1>(o, iters), mv = run<2>(o, iters), mix = run<3>(o, iters);
  printf("per iteration and SIMD (two waves): M  = 2 x %d MFMA             %.3f ms  (%.1f ns per MFMA)\n", NM, m, m * 1e6 / (2.0 * NM * iters));
  printf("                                    V  = 2 x %d v_fma           %.3f ms  (%.2f ns per v_fma)\n", NV, v, v * 1e6 / (2.0 * NV * iters));
  printf("                                    MV = %d MFMA | %d v_fma      %.3f ms  (sum of halves %.3f, max of halves %.3f)\n", NM, NV, mv, 0.5f * (m + v), 0.5f * (m > v ? m : v));
  printf("                                    MIX = 2 x (%d MFMA + %d v_fma interleaved 1 : %d)  %.3f ms  (sum %.3f, max %.3f)\n", NM, NV, NV / NM, mix, m + v, m > v ? m : v);
  return 0;
}
