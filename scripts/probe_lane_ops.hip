// probe_lane_ops.hip -- prints what v_permlane16_swap / v_permlane32_swap and the DPP controls used by the register epilogue of
// conv_mfma.hip do, lane by lane (the builtins' operand order is easy to get wrong and there is no GPU in the build container).
//   hipcc --offload-arch=gfx950 -O3 scripts/probe_lane_ops.hip -o scripts/bin/probe_lane_ops && scripts/bin/probe_lane_ops
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__global__ void probe(int* o) {
  const int l = threadIdx.x;
  unsigned a = l, b = 100 + l;
  auto s16 = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  auto s32 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  o[l] = s16[0]; o[64 + l] = s16[1]; o[128 + l] = s32[0]; o[192 + l] = s32[1];
  float r = (float)l;
  r = dpp_add<0xB1>(r); o[256 + l] = (int)r;
  r = dpp_add<0x4E>(r); o[320 + l] = (int)r;
  r = dpp_add<0x141>(r); o[384 + l] = (int)r;
  r = dpp_add<0x140>(r); o[448 + l] = (int)r;
  // the full reduction of the epilogue: A = 1000 + lane, B = 2000 + lane
  float A = 1000.f + l, B = 2000.f + l;
  auto sw = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, A), __builtin_bit_cast(unsigned, B), false, false);
  float q = __builtin_bit_cast(float, sw[0]) + __builtin_bit_cast(float, sw[1]);
  q = dpp_add<0xB1>(q); q = dpp_add<0x4E>(q); q = dpp_add<0x141>(q); q = dpp_add<0x140>(q);
  o[512 + l] = (int)q;
  // the same through inline asm (what conv_mfma.hip ships: the builtin + float add above is miscompiled by hipcc / ROCm 7.2)
  float A2 = 1000.f + l, B2 = 2000.f + l;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(A2), "+v"(B2));
  float q2 = A2 + B2;
  q2 = dpp_add<0xB1>(q2); q2 = dpp_add<0x4E>(q2); q2 = dpp_add<0x141>(q2); q2 = dpp_add<0x140>(q2);
  o[576 + l] = (int)q2;
}
int main() {
  int* d; hipMalloc(&d, 640 * 4);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  int h[640]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[10] = {"p16 ret0 (a')", "p16 ret1 (b')", "p32 ret0 (a')", "p32 ret1 (b')", "sum after quad_perm[1,0,3,2]", "+ quad_perm[2,3,0,1]", "+ row_half_mirror",
                          "+ row_mirror (expect row sums 120 376 632 888)", "full reduce via the BUILTIN (expect rows: 32496 64496 33520 65520 -- hipcc 7.2 gives 32240 64240 33264 65264: miscompiled)",
                          "full reduce via inline asm (expect rows: 32496 64496 33520 65520)"};
  for (int k = 0; k < 10; ++k) {
    printf("%s\n ", names[k]);
    for (int l = 0; l < 64; ++l) printf("%d%s", h[64 * k + l], l % 16 == 15 ? "\n " : " ");
    printf("\n");
  }
  return 0;
}
