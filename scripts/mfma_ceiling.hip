// MFMA issue-rate ceiling of the box under random operands (power-limited clock): register-resident operands, no LDS, no memory.
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_ceiling.hip -o scripts/bin/mfma_ceiling
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int KIND, int NVALU>
__global__ __launch_bounds__(512, 2) void k(const u32x4* __restrict__ src, float* out, int iters) {
  u32x4 a[4], b[2];
  for (int i = 0; i < 4; ++i) a[i] = src[(threadIdx.x + 512 * i) % 4096];
  for (int i = 0; i < 2; ++i) b[i] = src[(threadIdx.x + 512 * (4 + i)) % 4096];
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float v = threadIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        if (KIND == 0) acc[n * 2 + m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[n]), __builtin_bit_cast(bf16x8, b[m]), acc[n * 2 + m], 0, 0, 0);
        else acc[n * 2 + m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[n]), __builtin_bit_cast(f16x8, b[m]), acc[n * 2 + m], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < NVALU; ++q) v = __builtin_fmaf(v, 1.0001f, 0.5f);
      }
  }
  float s = v;
  for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  if (s == 123.456f) out[threadIdx.x] = s;
}

template <int KIND, int NVALU>
void run(const char* name, const u32x4* d, float* o, int blocks) {
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, NVALU>), dim3(blocks), dim3(512), 0, 0, d, o, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = 2.0 * 32 * 32 * 16 * 8.0 * iters * 8 * blocks;
    if (rep == 2) printf("%-34s blocks=%d  %.3f ms  %.1f TFLOP/s\n", name, blocks, ms, fl / ms / 1e9);
  }
}

int main() {
  std::vector<unsigned> h(4096 * 4);
  srand(1);
  auto fill = [&](int mode) {
    for (auto& x : h) {
      if (mode == 0) x = 0;
      else {  // two random bf16 / f16 values in [-1, 1): sign random, exponent around 0, random mantissa
        unsigned lo, hi;
        if (mode == 1) { lo = (rand() & 0x80ff) | ((0x7c + rand() % 3) << 7) | (rand() & 0x7f); hi = (rand() & 0x8000) | ((0x7c + rand() % 3) << 7) | (rand() & 0x7f); }
        else { lo = (rand() & 0x8000) | ((12 + rand() % 3) << 10) | (rand() & 0x3ff); hi = (rand() & 0x8000) | ((12 + rand() % 3) << 10) | (rand() & 0x3ff); }
        x = (lo & 0xffff) | (hi << 16);
      }
    }
  };
  u32x4* d; float* o;
  hipMalloc(&d, h.size() * 4); hipMalloc(&o, 4096);
  for (int mode = 0; mode < 3; ++mode) {
    fill(mode);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const char* mn = mode == 0 ? "zeros" : (mode == 1 ? "random bf16" : "random f16");
    char nm[128];
    if (mode != 2) { snprintf(nm, 128, "bf16 mfma, %s, 0 valu", mn); run<0, 0>(nm, d, o, 256); }
    if (mode != 1) { snprintf(nm, 128, "f16 mfma, %s, 0 valu", mn); run<1, 0>(nm, d, o, 256); }
    if (mode == 1) { snprintf(nm, 128, "bf16 mfma, %s, 2 valu/mfma", mn); run<0, 2>(nm, d, o, 256);
                     snprintf(nm, 128, "bf16 mfma, %s, 4 valu/mfma", mn); run<0, 4>(nm, d, o, 256);
                     snprintf(nm, 128, "bf16 mfma, %s, 6 valu/mfma", mn); run<0, 6>(nm, d, o, 256); }
    if (mode == 2) { snprintf(nm, 128, "f16 mfma, %s, 4 valu/mfma", mn); run<1, 4>(nm, d, o, 256);
                     snprintf(nm, 128, "f16 mfma, %s, 6 valu/mfma", mn); run<1, 6>(nm, d, o, 256); }
  }
  return 0;
}
