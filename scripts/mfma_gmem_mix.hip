// Would the direct conv kernel gain from taking its WEIGHT fragments straight from L2 (global_load_dwordx4 into registers, a few
// iterations ahead) instead of through the LDS ring?  Same loop as mfma_lds_mix.hip (8 MFMAs 32x32x16 bf16 per iteration, random
// operands, NV VALU ops per MFMA, 512 threads x 256 blocks, 2 waves / SIMD), with NL operand reads per iteration from LDS and NG from a
// 1.2 MB global buffer that every block streams in the same order (= one layer's weights: L2 / MALL resident); waves w and w + 4 of a
// block read the same addresses (two pixel groups share a cout group).  Global reads are DIST iterations ahead (register ring).
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_gmem_mix.hip -o scripts/bin/mfma_gmem_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
constexpr int GBUF = 73728;   // u32x4 elements = 1.18 MB (256 x 256 x 9 bf16)

template <int NL, int NG, int NV>
__global__ __launch_bounds__(512, 2) void k(const u32x4* __restrict__ src, const u32x4* __restrict__ gw, float* out, int iters) {
  __shared__ u32x4 lds[4096];   // 64 KiB of random bf16 pairs
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = src[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, grp = (threadIdx.x >> 6) & 3;
  u32x4 op[6];
  for (int i = 0; i < 6; ++i) op[i] = lds[(threadIdx.x + 64 * i) & 4095];
  constexpr int NGR = NG > 0 ? NG : 1;
  u32x4 ring[3][NGR];
  unsigned goff = (unsigned)(grp * 64 + lane);
  auto gload = [&](u32x4 (&dst)[NGR]) {
#pragma unroll
    for (int r = 0; r < NG; ++r) { dst[r] = gw[goff]; goff += 256; if (goff >= GBUF) goff -= GBUF; }
  };
  if (NG > 0) { gload(ring[0]); gload(ring[1]); }
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float v = threadIdx.x * 1e-3f;
  int idx = threadIdx.x;
  auto body = [&](u32x4 (&cur)[NGR], u32x4 (&far)[NGR]) {
    if (NG > 0) gload(far);               // two iterations ahead
    u32x4 nl[6];
#pragma unroll
    for (int r = 0; r < NL; ++r) nl[r] = lds[(idx + 67 * r) & 4095];
    idx += 193;
    // operands: the first NG come from the global ring, the rest from LDS (one iteration ahead)
    u32x4 a[2], b[4];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const u32x4 x = i < NG ? cur[i] : op[i];
      if (i < 2) a[i] = x; else b[i - 2] = x;
    }
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        acc[n * 4 + m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[n]), __builtin_bit_cast(bf16x8, b[m]), acc[n * 4 + m], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < NV; ++q) v = __builtin_fmaf(v, 1.0001f, 0.5f);
      }
#pragma unroll
    for (int r = 0; r < NL; ++r) op[NG + r < 6 ? NG + r : 5] = nl[r];
  };
  for (int it = 0; it < iters; it += 3) {
    body(ring[0], ring[2]);
    body(ring[1], ring[0]);
    body(ring[2], ring[1]);
  }
  float s = v;
  for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  if (s == 123.456f) out[threadIdx.x] = s;
}

template <int NL, int NG, int NV>
void run(const u32x4* d, const u32x4* g, float* o) {
  const int iters = 19998, blocks = 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NL, NG, NV>), dim3(blocks), dim3(512), 0, 0, d, g, o, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double fl = 2.0 * 32 * 32 * 16 * 8.0 * iters * 8 * blocks;
  printf("LDS reads/MFMA %.2f  global reads/MFMA %.2f  valu/MFMA %d  ->  %.3f ms  %.1f TFLOP/s\n", NL / 8.0, NG / 8.0, NV, ms, fl / ms / 1e9);
}

int main() {
  std::vector<unsigned> h((size_t)GBUF * 4);
  srand(1);
  for (auto& x : h) {
    unsigned lo = (rand() & 0x8000) | ((0x7c + rand() % 3) << 7) | (rand() & 0x7f), hi = (rand() & 0x8000) | ((0x7c + rand() % 3) << 7) | (rand() & 0x7f);
    x = (lo & 0xffff) | (hi << 16);
  }
  u32x4 *d, *g; float* o;
  hipMalloc(&d, 4096 * 16); hipMalloc(&g, h.size() * 4); hipMalloc(&o, 4096);
  hipMemcpy(d, h.data(), 4096 * 16, hipMemcpyHostToDevice);
  hipMemcpy(g, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  run<6, 0, 3>(d, g, o);   // today's kernel: 4 weight + 2 pixel fragments per 8 MFMAs, all from LDS
  run<4, 0, 3>(d, g, o);
  run<2, 0, 3>(d, g, o);
  run<4, 2, 3>(d, g, o);   // 2 weight fragments (64 couts per wave) from L2, 4 pixel fragments from LDS
  run<2, 4, 3>(d, g, o);   // 4 weight fragments (128 couts per wave) from L2, 2 pixel fragments from LDS
  run<6, 0, 0>(d, g, o);
  run<4, 2, 0>(d, g, o);
  run<2, 4, 0>(d, g, o);
  return 0;
}
