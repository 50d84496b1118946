// Issue rate of the legacy v_mfma_f32_32x32x8_bf16 (4 bf16 per lane) next to v_mfma_f32_32x32x16_bf16 and v_mfma_f32_32x32x2_f32 on gfx950:
// register-resident operands, 2 x 8 waves per CU.   hipcc --offload-arch=gfx950 -O3 scripts/mfma_x8_rate.hip -o scripts/bin/mfma_x8_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int KIND>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters) {
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  const unsigned h = threadIdx.x * 2654435761u;
  bf16x8 a8, b8; s16x4 a4, b4;
  for (int e = 0; e < 8; ++e) { a8[e] = (__bf16)(float)((h >> e) & 7) * 0.25f; b8[e] = (__bf16)(float)((h >> (e + 3)) & 7) * 0.125f; }
  for (int e = 0; e < 4; ++e) { a4[e] = (short)(0x3f00 | ((h >> e) & 0x7f)); b4[e] = (short)(0x3e80 | ((h >> (e + 5)) & 0x7f)); }
  const float af = (float)(h & 15) * 0.1f, bf = (float)((h >> 4) & 15) * 0.05f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[i], 0, 0, 0);
      else if (KIND == 1) acc[i] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a4, b4, acc[i], 0, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  if (s == 123.456f) out[threadIdx.x] = s;
}
template <int KIND>
void run(const char* name, double flop_per_mfma, float* o, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND>), dim3(512), dim3(512), 0, 0, o, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double n = 8.0 * iters * 8 * 512;   // MFMAs
  printf("%-28s %.3f ms  %.1f TFLOP/s  %.2f ns per MFMA and SIMD\n", name, ms, flop_per_mfma * n / ms / 1e9, ms * 1e6 / (n / 1024.0));
}
int main() {
  float* o; hipMalloc(&o, 4096);
  run<0>("v_mfma_f32_32x32x16_bf16", 2.0 * 32 * 32 * 16, o, 20000);
  run<1>("v_mfma_f32_32x32x8_bf16", 2.0 * 32 * 32 * 8, o, 20000);
  run<2>("v_mfma_f32_32x32x2_f32", 2.0 * 32 * 32 * 2, o, 5000);
  return 0;
}
