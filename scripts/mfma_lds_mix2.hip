// mfma_lds_mix.hip with the operand reads TWO iterations ahead of the MFMAs that consume them (three register sets, the loop
// unrolled by 3 so that the rotation is compile-time): does a deeper fragment pipeline lift the 0.75-reads-per-MFMA mix above
// the 1525 TFLOP/s of the one-iteration pipeline?   D = 1 reproduces mfma_lds_mix.hip.
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_lds_mix2.hip -o scripts/bin/mfma_lds_mix2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int NR, int NV, int D, bool FENCE>
__global__ __launch_bounds__(512, 2) void k(const u32x4* __restrict__ src, float* out, int iters) {
  __shared__ u32x4 lds[4096];   // 64 KiB of random bf16 pairs
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = src[i];
  __syncthreads();
  u32x4 a[3][2], b[3][4];
  for (int s = 0; s < 3; ++s) {
    for (int i = 0; i < 2; ++i) a[s][i] = lds[(threadIdx.x + 64 * i + 7 * s) & 4095];
    for (int i = 0; i < 4; ++i) b[s][i] = lds[(threadIdx.x + 512 + 64 * i + 11 * s) & 4095];
  }
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float v = threadIdx.x * 1e-3f;
  int idx = threadIdx.x;
  for (int it = 0; it < iters; it += 3) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int cs = u, rs = (u + D) % 3;   // compute set / set being filled (consumed D iterations later)
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        if (r < NR) {
          const u32x4 x = lds[(idx + 67 * r) & 4095];
          if (r < 2) a[rs][r] = x; else if (r < 6) b[rs][r - 2] = x;
          else asm volatile("" :: "v"(x));
        }
      }
      idx += 193;
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          acc[n * 4 + m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[cs][n]), __builtin_bit_cast(bf16x8, b[cs][m]), acc[n * 4 + m], 0, 0, 0);
#pragma unroll
          for (int q = 0; q < NV; ++q) v = __builtin_fmaf(v, 1.0001f, 0.5f);
        }
      if (FENCE) __builtin_amdgcn_sched_barrier(0);   // keep hipcc from sinking the reads towards their first use
    }
  }
  float s = v;
  for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  if (s == 123.456f) out[threadIdx.x] = s;
}

template <int NR, int NV, int D, bool FENCE>
void run(const u32x4* d, float* o) {
  const int iters = 19998, blocks = 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NR, NV, D, FENCE>), dim3(blocks), dim3(512), 0, 0, d, o, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double fl = 2.0 * 32 * 32 * 16 * 8.0 * iters * 8 * blocks;
  printf("reads/MFMA %.2f  valu/MFMA %d  distance %d %s ->  %.3f ms  %.1f TFLOP/s\n", NR / 8.0, NV, D, FENCE ? "fenced" : "free  ", ms, fl / ms / 1e9);
}

int main() {
  std::vector<unsigned> h(4096 * 4);
  srand(1);
  for (auto& x : h) {
    unsigned lo = (rand() & 0x8000) | ((0x7c + rand() % 3) << 7) | (rand() & 0x7f), hi = (rand() & 0x8000) | ((0x7c + rand() % 3) << 7) | (rand() & 0x7f);
    x = (lo & 0xffff) | (hi << 16);
  }
  u32x4* d; float* o;
  hipMalloc(&d, h.size() * 4); hipMalloc(&o, 4096);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  run<0, 0, 1, false>(d, o);
  run<6, 0, 1, false>(d, o); run<6, 0, 1, true>(d, o); run<6, 0, 2, false>(d, o); run<6, 0, 2, true>(d, o);
  run<6, 3, 1, false>(d, o); run<6, 3, 1, true>(d, o); run<6, 3, 2, false>(d, o); run<6, 3, 2, true>(d, o);
  run<8, 0, 1, true>(d, o); run<8, 0, 2, true>(d, o);
  return 0;
}
