// calib.hip -- box calibration for the bench line (round 6).
//
// The MI355X boxes of the pool differ by +-2.5 % on every MFMA-heavy kernel, and the chip runs at its POWER limit, not at its clock
// limit, while the convolutions execute (MEASUREMENTS.md R5.1: 2.10 GHz at 1300 W of a 2.4 GHz part).  A bench line alone therefore
// cannot tell a code change from a different box.  fd_calibrate_mfma runs a fixed matrix-core issue loop -- register-resident
// operands, random bf16 bit patterns (operand toggling is what draws the power), no LDS, no memory traffic, two workgroups of 8 waves
// per compute unit -- and reports the rate the box sustains on it.  bench.py emits it next to `value` (`box_calibration`), so that
// `value / mfma_tflops` can be compared across boxes and rounds.  It is the loop of scripts/mfma_ceiling.hip (round 2: 1850 TFLOP/s
// on random bf16, 2412 on zeros) as a library entry point; it has no counterpart in the reference (enhance.py:120-136 times a call).
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// a bf16 pair with random signs and mantissas and exponents around 2^-1: the operand statistics of normalised activations
__device__ __forceinline__ unsigned rand_bf16_pair(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  const unsigned lo = (x & 0x807fu) | ((0x7cu + (x >> 7) % 3u) << 7);
  const unsigned y = x * 0x9e3779b9u + 0x85ebca6bu;
  const unsigned hi = (y & 0x807fu) | ((0x7cu + (y >> 7) % 3u) << 7);
  return lo | (hi << 16);
}

__global__ __launch_bounds__(512, 2) void calib_mfma_kernel(float* __restrict__ out, int iters, unsigned seed) {
  u32x4 a[4], b[2];
  const unsigned id = (blockIdx.x * 512u + threadIdx.x) * 24u + seed;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) a[i][e] = rand_bf16_pair(id + i * 4 + e);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) b[i][e] = rand_bf16_pair(id + 16 + i * 4 + e);
  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int m = 0; m < 2; ++m)
        acc[n * 2 + m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[n]), __builtin_bit_cast(bf16x8, b[m]), acc[n * 2 + m], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  if (s == 123.456f) out[threadIdx.x] = s;   // (keeps the loop alive; never true for these operands)
}

}  // namespace

// Runs `repeats` timed launches of the issue loop (after one untimed launch) on `stream` and returns the mean rate in TFLOP/s and the
// total time of the timed launches.  iters <= 0: 100000 iterations (about 50 ms per launch at the power-limited clock).  Synchronises
// the stream.  `scratch`: 2 KiB of device memory (never written in practice).
extern "C" int fd_calibrate_mfma(float* scratch, int iters, int repeats, double* tflops, double* ms_total, void* stream) {
  FD_REQUIRE(scratch && tflops && repeats >= 1 && repeats <= 64, "fd_calibrate_mfma: bad arguments");
  if (iters <= 0) iters = 100000;
  int dev = 0, cus = 0;
  FD_HIP(hipGetDevice(&dev));
  FD_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const int blocks = 2 * cus;
  hipStream_t st = fd_stream(stream);
  hipEvent_t e0, e1;
  FD_HIP(hipEventCreate(&e0));
  FD_HIP(hipEventCreate(&e1));
  hipLaunchKernelGGL(calib_mfma_kernel, dim3(blocks), dim3(512), 0, st, scratch, iters / 4 + 1, 1u);   // clocks / power ramp
  FD_HIP(hipEventRecord(e0, st));
  for (int r = 0; r < repeats; ++r) hipLaunchKernelGGL(calib_mfma_kernel, dim3(blocks), dim3(512), 0, st, scratch, iters, 7u + r);
  FD_HIP(hipEventRecord(e1, st));
  FD_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  FD_HIP(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  FD_LAUNCH_CHECK();
  const double flop = 2.0 * 32 * 32 * 16 * 8.0 /* MFMAs per iteration and wave */ * 8 /* waves */ * (double)iters * blocks * repeats;
  *tflops = ms > 0.f ? flop / (ms * 1e-3) / 1e12 : 0.0;
  if (ms_total) *ms_total = ms;
  return FD_OK;
}
