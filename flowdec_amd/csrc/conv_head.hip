// conv_head.hip -- the 3x3 pyramid-head convolutions of NCSN++ (C -> 4 channels; ncsnpp.py:352-362: GroupNorm + SiLU +
// conv3x3 on every up level, `pyramid = upsample(pyramid) + head(h)`), bf16 mode.
//
// With Cout = 4 a (chunk, tap) step of the generic implicit-GEMM kernel (conv_mfma.hip) is 4 MFMAs per wave: the launch is
// bound by the memory round trip of every 32-channel chunk (one 64-byte slice per halo pixel: 0.47 ms for a 29 GFLOP launch at
// full resolution, 805 MB at 1.7 TB/s), not by its arithmetic.  This kernel keeps the same tile (16 x 16 pixels, 4 waves) and
//   * holds ALL weights of the layer in LDS: only rows 0..3 of every [32 x 64 B] slab of the generic packing are real, so a
//     256 -> 4 layer is 72 x 256 B = 18 KiB, loaded once per workgroup; rows 4..31 of the MFMA A operand come from one shared
//     zero row (an LDS broadcast): no weight ring, no DMA waits, 2 barriers per chunk, both only for the halo buffer;
//   * keeps TWO chunks of halo in flight in registers (the loads of chunk c+3 are issued when chunk c+1 is stored), so that a
//     chunk's global loads have two chunk times to land;
//   * ONE halo buffer -> 47 KiB (128 input channels: three workgroups per CU) / 56 KiB (256: two) of LDS;
//   * stores straight from registers: the 4 output channels of a pixel are accumulator registers 0..3 of one lane.
// Contract = fd_conv2d with Cout = 4, ksize 3, an even number of 32-channel chunks, no folded shortcut, no statistics output.
#include <type_traits>

#include "conv_common.h"

namespace {

using namespace fdconv;

constexpr int NTH = 256, TH = 16, TW = 16, HH = TH + 2, HW = TW + 2;
constexpr int HALO_BYTES = HH * PITCH * ROWB;      // one buffer, the direct kernel's geometry (PITCH 24, ROWB 80)
constexpr int MAX_STEPS = 9 * 16;                  // up to 512 input channels
constexpr int ZERO_OFF = HALO_BYTES;               // 64 B of zeros: rows 4..31 of the A operand
constexpr int AFFH_OFF = ZERO_OFF + 64;
constexpr int W_OFF = AFFH_OFF + AFF_BYTES;        // [step][4 rows][64 B], LAST: a launch allocates what its layer needs
constexpr int LDS_MAX = W_OFF + MAX_STEPS * 256;   // 34560 + 64 + 4096 + 36864 = 75584 (512 input channels)
__host__ __device__ constexpr int lds_bytes(int nsteps) { return W_OFF + nsteps * 256; }   // 256 channels: 57152 -> 2 workgroups per CU; 128: 47936 -> 3
constexpr int PPP = NTH / 4;
constexpr int HITER = (HH * HW + PPP - 1) / PPP;   // 6
constexpr int CK = 32, EPS = 8;

// silu(a*x+d) on the 8 bf16 channels of one 16-byte slot ((a, d) pairs from the LDS table) -- as in conv_mfma.hip
__device__ __forceinline__ u32x4 act_slot(u32x4 raw, const char* ad) {
  u32x4 out;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(ad + 16 * j);
    const unsigned u = raw[j];
    const float x0 = __builtin_bit_cast(float, u << 16), x1 = __builtin_bit_cast(float, u & 0xffff0000u);
    bf16x2 r = {(bf16)fd_silu(fmaf(x0, a[0], a[1])), (bf16)fd_silu(fmaf(x1, a[2], a[3]))};
    out[j] = __builtin_bit_cast(unsigned, r);
  }
  return out;
}

// one chunk of halo in registers: raw data + what its conversion needs
struct HaloRegs {
  u32x4 v[HITER];
  unsigned mask;   // bit i: slot i holds real data (pixel inside the image, channel < C)
  int aff;         // byte offset of the slot's (a, d) pairs in the LDS table
};

template <bool ACT, bool SKIP>
__global__ __launch_bounds__(NTH, 2) void conv_head_kernel(ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const hbuf = smem;
  char* const afftab = smem + AFFH_OFF;

  const int bid = blockIdx.x, nblk = gridDim.x;
  int lid;
  {
    const int xcd = bid & 7, qq = nblk >> 3, rr = nblk & 7;
    lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  }
  const int tw_i = lid % p.tiles_w;
  const int th_i = (lid / p.tiles_w) % p.tiles_h;
  const int b = lid / (p.tiles_w * p.tiles_h);
  const int h0 = th_i * TH, w0 = tw_i * TW;
  const int H = p.H, W = p.W;
  const int t = threadIdx.x, q = t & 3, prow = t >> 2;

  // ---- halo loader (duplicate-slot trick: no divergent branch in the conversion) ------------------------------------------
  int pixl[HITER], hlds[HITER];
  unsigned pvalid = 0;
#pragma unroll
  for (int i = 0; i < HITER; ++i) {
    int hp = prow + i * PPP;
    if (hp >= HH * HW) hp -= PPP;
    const int hr = hp / HW, hc = hp - hr * HW;
    const int gh = h0 - 1 + hr, gw = w0 - 1 + hc;
    const bool ok = gh >= 0 && gh < H && gw >= 0 && gw < W;
    pixl[i] = ok ? gh * W + gw : 0;
    hlds[i] = (hr * PITCH + hc) * ROWB + q * 16;
    if (ok) pvalid |= 1u << i;
  }
  const size_t img_elems = (size_t)H * W;
  int nchunks = 0;
  for (int s = 0; s < p.nseg; ++s) nchunks += (p.seg[s].C + CK - 1) / CK;
  const int nsteps = nchunks * 9;

  int cs = 0, cch = -1;
  bool cur_end = false;
  // advance the load cursor by one chunk and issue its halo loads into `r` (past the end: a harmless re-read of element 0)
  auto load_next = [&](HaloRegs& r) {
    if (!cur_end) {
      ++cch;
      if (cch >= (p.seg[cs].C + CK - 1) / CK) { ++cs; cch = 0; }
      if (cs >= p.nseg) { cur_end = true; cs = p.nseg - 1; }
    }
    const Seg sg = p.seg[cs];
    const bf16* src = reinterpret_cast<const bf16*>(sg.src) + (size_t)b * img_elems * sg.C;
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(src), 0, (int)(img_elems * sg.C * sizeof(bf16)), 0x00020000);
    const int c = cch * CK + q * EPS;
    const bool ok = c < sg.C && !cur_end;
    const int nc = ok ? c : 0;
    r.aff = sg.aff_off >= 0 ? (sg.aff_off + nc) * 8 : 0;
    r.mask = ok ? pvalid : 0u;
    const int on = cur_end ? 0 : 1;
#pragma unroll
    for (int i = 0; i < HITER; ++i) r.v[i] = __builtin_amdgcn_raw_buffer_load_b128(srd, (pixl[i] * sg.C + nc) * 2 * on, 0, 0);
  };
  // in registers: silu(a*x+d), zero padding AFTER the activation (AND with a lane mask, no select)
  auto convert_slot = [&](HaloRegs& r, int i) {
    u32x4 v = r.v[i];
    if constexpr (ACT) v = act_slot(v, afftab + r.aff);
    const unsigned m = ((r.mask >> i) & 1u) ? 0xffffffffu : 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] &= m;
    r.v[i] = v;
  };
  auto store_halo = [&](const HaloRegs& r) {
#pragma unroll
    for (int i = 0; i < HITER; ++i) *reinterpret_cast<u32x4*>(hbuf + hlds[i]) = r.v[i];
  };

  // ---- prologue: affine table, all weights (rows 0..3 of every step slab of the generic packing), zero row, first halo ----
  HaloRegs hA, hB;
  load_next(hA);   // chunk 0
  load_next(hB);   // chunk 1
  if (ACT) {
    const float* ap = p.affine + (size_t)b * p.affC * 2;
    for (int i = t; i < p.affC / 2; i += NTH) *reinterpret_cast<f32x4*>(afftab + i * 16) = *reinterpret_cast<const f32x4*>(ap + i * 4);
  }
  for (int i = t; i < nsteps * 16; i += NTH) {   // 16 x 16-byte pieces per step
    const int st = i >> 4, piece = i & 15;
    *reinterpret_cast<u32x4*>(smem + W_OFF + st * 256 + piece * 16) =
        *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.w) + ((size_t)st * p.CoutPad) * WROWB + piece * 16);
  }
  if (t < 4) *reinterpret_cast<u32x4*>(smem + ZERO_OFF + t * 16) = u32x4{0u, 0u, 0u, 0u};
  __syncthreads();
#pragma unroll
  for (int i = 0; i < HITER; ++i) convert_slot(hA, i);
  store_halo(hA);
  load_next(hA);   // chunk 2
  __syncthreads();

  // ---- per-lane fragment coordinates: wave = two 4x8-pixel patches; A operand row = cout (rows >= 4: the zero row) ---------
  const int lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  int pbase[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int pi = wave * 2 + mi;
    const int r = 4 * (pi >> 1) + (l31 >> 3), c = 8 * (pi & 1) + (l31 & 7);
    pbase[mi] = (r * PITCH + c) * ROWB + lh * 16;
  }
  const bool wrow = l31 < 4;
  const int wlane = W_OFF + l31 * WROWB + lh * 16;   // + step * 256 + ks * 32
  const int zlane = ZERO_OFF + lh * 16;

  f32x16 acc[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[mi][e] = 0.f;

  auto lds_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  // 18 (tap, k-half) steps per chunk, each 1 weight + 2 pixel fragment reads and 2 MFMAs; the reads run one step ahead of the
  // MFMAs (explicit register double buffer); the halo of the next chunk (in `nx`, loaded two chunks ago) is converted one slot
  // per tap (taps 3..8) so that its VALU work sits between the MFMAs.
  auto read_step = [&](u32x4& wf, u32x4 (&pf)[2], int wch, int st) {   // st = 2 * tap + ks (compile time after unrolling)
    const int tap = st >> 1, ks = st & 1;
    const int imm = ((tap / 3) * PITCH + (tap % 3)) * ROWB;
    wf = *reinterpret_cast<const u32x4*>(smem + (wrow ? wlane + wch + tap * 256 + ks * 32 : zlane));
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) pf[mi] = *reinterpret_cast<const u32x4*>(hbuf + pbase[mi] + imm + 32 * ks);
  };
  u32x4 wfA, wfB, pfA[2], pfB[2];
  auto chunk = [&](int ch, HaloRegs& nx) {
    const int wch = ch * 9 * 256;
    read_step(wfA, pfA, wch, 0);
#pragma unroll
    for (int st = 0; st < 18; st += 2) {
      read_step(wfB, pfB, wch, st + 1);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wfA), __builtin_bit_cast(bf16x8, pfA[mi]), acc[mi], 0, 0, 0);
      if (st / 2 >= 3) convert_slot(nx, st / 2 - 3);
      if (st + 2 < 18) read_step(wfA, pfA, wch, st + 2);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wfB), __builtin_bit_cast(bf16x8, pfB[mi]), acc[mi], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // 2 DS read
        __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);  // 16 VALU
      }
    }
    lds_barrier();      // every read of this chunk's halo is done
    store_halo(nx);     // chunk ch + 1
    load_next(nx);      // chunk ch + 3: two chunk times to land
    lds_barrier();      // next halo published
  };
  for (int ch = 0; ch < nchunks; ch += 2) {   // (even chunk count: the two register sets alternate at compile time)
    chunk(ch, hB);
    chunk(ch + 1, hA);
  }

  // ---- epilogue: lanes 0..31 hold the 4 couts of their pixel in acc[mi][0..3] ---------------------------------------------
  if (lh == 0) {
    bf16* out = reinterpret_cast<bf16*>(p.out);
    const bf16* skip = reinterpret_cast<const bf16*>(p.skip);
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
      const float* bp = p.bias + (size_t)(p.bias_rows > 1 ? b : 0) * p.Cout;
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[j] = bp[j];
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int pi = wave * 2 + mi;
      const int gh = h0 + 4 * (pi >> 1) + (l31 >> 3), gw = w0 + 8 * (pi & 1) + (l31 & 7);
      if (gh < H && gw < W) {
        const size_t o = (((size_t)b * H + gh) * W + gw) * 4;
        float v[4] = {acc[mi][0], acc[mi][1], acc[mi][2], acc[mi][3]};
        if constexpr (SKIP) {
          const bf16x4 s = *reinterpret_cast<const bf16x4*>(skip + o);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += (float)s[j];
        }
        bf16x4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = (bf16)((v[j] + bv[j]) * p.scale);
        *reinterpret_cast<bf16x4*>(out + o) = r;
      }
    }
  }
}

}  // namespace

bool fd_head_supported(const ConvArgs& a, int ksize, int dtype) {
  if (!(dtype == FD_BF16 && ksize == 3 && a.Cout == 4 && a.stats == nullptr)) return false;
  int chunks = 0;
  for (int s = 0; s < a.nseg; ++s) {
    if (a.seg[s].taps != 9) return false;
    if ((a.seg[s].aff_off >= 0) != (a.affine != nullptr)) return false;
    chunks += (a.seg[s].C + CK - 1) / CK;
  }
  return chunks % 2 == 0 && chunks * 9 <= MAX_STEPS;
}

int fd_head_init_attributes() {
  FD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_head_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX));
  FD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_head_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX));
  FD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_head_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX));
  FD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_head_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX));
  return FD_OK;
}

int fd_head_launch(ConvArgs a, hipStream_t st) {
  a.tiles_h = fd_cdiv(a.H, TH);
  a.tiles_w = fd_cdiv(a.W, TW);
  a.tiles_n = 1;
  const long long nblk = (long long)a.B * a.tiles_h * a.tiles_w;
  FD_REQUIRE(nblk > 0 && nblk < (1ll << 31), "conv grid out of range");
  const dim3 grid((unsigned)nblk), block(NTH);
  int chunks = 0;
  for (int s = 0; s < a.nseg; ++s) chunks += (a.seg[s].C + CK - 1) / CK;
  const int LDS_BYTES = lds_bytes(chunks * 9);
  if (a.affine) {
    if (a.skip) hipLaunchKernelGGL((conv_head_kernel<true, true>), grid, block, LDS_BYTES, st, a);
    else hipLaunchKernelGGL((conv_head_kernel<true, false>), grid, block, LDS_BYTES, st, a);
  } else {
    if (a.skip) hipLaunchKernelGGL((conv_head_kernel<false, true>), grid, block, LDS_BYTES, st, a);
    else hipLaunchKernelGGL((conv_head_kernel<false, false>), grid, block, LDS_BYTES, st, a);
  }
  FD_LAUNCH_CHECK();
  return FD_OK;
}
