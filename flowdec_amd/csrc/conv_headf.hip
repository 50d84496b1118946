// conv_headf.hip -- the 3x3 pyramid-head convolutions of NCSN++ (C -> 4 channels; ncsnpp.py:352-362: GroupNorm + SiLU + conv3x3 on every
// up level, `pyramid = upsample(pyramid) + head(h)`) in EXACT FLOAT32 (precision 'fp32'; round 6).
//
// The generic float32 kernel (conv_mfma.hip) pads the 4 output channels to the 32 rows of v_mfma_f32_32x32x2_f32: 7 of 8 MFMAs multiply
// zeros, and the f32 MFMA is the slow one (0.9 ms per head, 2.9 % of an fp32 step once the 2-D Winograd kernel had taken the rest).  This
// kernel uses the one matrix instruction whose M IS 4: v_mfma_f32_4x4x1_16B_f32 = 16 independent 4 x 4 outer products per instruction:
//   A (4 x 1 per block) = the 4 couts' weights of one input channel (lane & 3 = cout: the same four values in all 16 blocks),
//   B (1 x 4 per block) = that channel of 4 pixels per block = one pixel per lane,   D: lane (block, j) holds the 4 couts of ITS pixel.
// No padding anywhere: a (tap, 4-channel) step is one 16-byte weight read, one 16-byte activation read and 4 MFMAs per wave of 64 pixels.
//   * one workgroup = 16 x 16 output pixels, 4 waves (rows 4 w .. 4 w + 3 each);
//   * all weights of the layer in LDS (rows 0 .. 3 of every [CoutPad x 64 B] slab of the generic float32 packing: 256 B per (chunk, tap)),
//     the GroupNorm affine table next to them; LDS is allocated per layer: 50 KiB at 128 input channels (three workgroups per CU), 68 KiB at 256;
//   * K walks 16-channel chunks through ONE halo buffer (18 rows of 1536 B: 18 pixels x 80 B, padded so that the 16-byte reads of a wave
//     are free of bank conflicts at every tap), two chunks of raw halo in flight in registers, activated ([silu(a x + d)], zero padding)
//     between the MFMAs of the chunk before -- the structure of conv_head.hip.
// Contract = fd_conv2d with FD_F32 storage, Cout = 4, ksize 3, an even number of 16-channel chunks, no folded shortcut, no statistics.
#include <type_traits>

#include "conv_common.h"

namespace {

using namespace fdconv;

constexpr int NTH = 256, TH = 16, TW = 16, HH = TH + 2, HW = TW + 2;
constexpr int CK = 16;                              // channels per chunk (one 64-byte row of the generic float32 packing)
constexpr int PIXB = 80;                            // bytes per halo pixel: 64 B of data + 16 B pad
constexpr int HROWB = 1536;                         // halo row pitch: 18 x 80 = 1440 -> 96 sixteen-byte units (0 mod 16: see header)
constexpr int HALO_BYTES = HH * HROWB;              // 27648
constexpr int AFFF_OFF = HALO_BYTES;
constexpr int WF_OFF = AFFF_OFF + AFF_BYTES;        // [step][4 couts][64 B], LAST: a launch allocates what its layer needs
constexpr int MAX_STEPS = 9 * 32;                   // up to 512 input channels
__host__ __device__ constexpr int lds_bytes(int nsteps) { return WF_OFF + nsteps * 256; }
constexpr int PPP = NTH / 4;                        // halo pixels per pass (4 lanes = the four 16-byte slots of a pixel)
constexpr int HITER = (HH * HW + PPP - 1) / PPP;    // 6

struct HaloRegs {
  f32x4 v[HITER];
  unsigned mask;   // bit i: slot i holds real data (pixel inside the image, channel < C)
  int aff;         // byte offset of the slot's (a, d) pairs in the LDS table
};

template <bool ACT, bool SKIP>
__global__ __launch_bounds__(NTH, 2) void conv_headf_kernel(ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const hbuf = smem;
  char* const afftab = smem + AFFF_OFF;

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
    hlds[i] = hr * HROWB + hc * PIXB + q * 16;
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
    const float* src = reinterpret_cast<const float*>(sg.src) + (size_t)b * img_elems * sg.C;
    const int c = cch * CK + q * 4;
    const bool ok = c < sg.C && !cur_end;
    const int nc = ok ? c : 0;
    r.aff = sg.aff_off >= 0 ? (sg.aff_off + nc) * 8 : 0;
    r.mask = ok ? pvalid : 0u;
    const int on = cur_end ? 0 : 1;
#pragma unroll
    for (int i = 0; i < HITER; ++i) r.v[i] = *reinterpret_cast<const f32x4*>(src + ((size_t)pixl[i] * sg.C + nc) * on);
  };
  // in registers: silu(a*x+d), zero padding AFTER the activation
  auto convert_slot = [&](HaloRegs& r, int i) {
    f32x4 v = r.v[i];
    if constexpr (ACT) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(afftab + r.aff), a1 = *reinterpret_cast<const f32x4*>(afftab + r.aff + 16);
      v[0] = fd_silu(fmaf(v[0], a0[0], a0[1]));
      v[1] = fd_silu(fmaf(v[1], a0[2], a0[3]));
      v[2] = fd_silu(fmaf(v[2], a1[0], a1[1]));
      v[3] = fd_silu(fmaf(v[3], a1[2], a1[3]));
    }
    if (!((r.mask >> i) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
    r.v[i] = v;
  };
  auto store_halo = [&](const HaloRegs& r) {
#pragma unroll
    for (int i = 0; i < HITER; ++i) *reinterpret_cast<f32x4*>(hbuf + hlds[i]) = r.v[i];
  };

  // ---- prologue: affine table, all weights (rows 0..3 of every step slab of the generic packing), first halo ----------------
  HaloRegs hA, hB;
  load_next(hA);   // chunk 0
  load_next(hB);   // chunk 1
  if (ACT) {
    const float* ap = p.affine + (size_t)b * p.affC * 2;
    for (int i = t; i < p.affC / 2; i += NTH) *reinterpret_cast<f32x4*>(afftab + i * 16) = *reinterpret_cast<const f32x4*>(ap + i * 4);
  }
  for (int i = t; i < nsteps * 16; i += NTH) {   // 16 x 16-byte pieces per step
    const int st = i >> 4, piece = i & 15;
    *reinterpret_cast<f32x4*>(smem + WF_OFF + st * 256 + piece * 16) =
        *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(p.w) + ((size_t)st * p.CoutPad) * WROWB + piece * 16);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < HITER; ++i) convert_slot(hA, i);
  store_halo(hA);
  load_next(hA);   // chunk 2
  __syncthreads();

  // ---- per-lane coordinates: wave w = output rows 4 w .. 4 w + 3, lane = (row lane >> 4, column lane & 15); A operand row = cout lane & 3
  const int lane = t & 63, wave = t >> 6;
  const int pbase = ((wave * 4 + (lane >> 4)) * HROWB + (lane & 15) * PIXB);
  const int wlane = WF_OFF + (lane & 3) * WROWB;   // + step * 256 + q * 16

  f32x4 acc[2];   // two accumulators: consecutive MFMAs never wait for each other
  acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
  acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto lds_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  // 36 (tap, channel quad) steps per chunk, each one weight read, one activation read and 4 MFMAs; the halo of the next chunk (in `nx`,
  // loaded two chunks ago) is converted one slot per tap (taps 3..8) so that its vector work sits between the MFMAs.
  auto chunk = [&](int ch, HaloRegs& nx) {
    const int wch = ch * 9 * 256;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int imm = (tap / 3) * HROWB + (tap % 3) * PIXB;
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(smem + wlane + wch + tap * 256 + qq * 16);
        const f32x4 xv = *reinterpret_cast<const f32x4*>(hbuf + pbase + imm + qq * 16);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k & 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[k], xv[k], acc[k & 1], 0, 0, 0);
      }
      if (tap >= 3) convert_slot(nx, tap - 3);
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

  // ---- epilogue: every lane holds the 4 couts of its pixel ------------------------------------------------------------------
  {
    float* out = reinterpret_cast<float*>(p.out);
    const float* skip = reinterpret_cast<const float*>(p.skip);
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + (size_t)(p.bias_rows > 1 ? b : 0) * p.Cout);
    const int gh = h0 + wave * 4 + (lane >> 4), gw = w0 + (lane & 15);
    if (gh < H && gw < W) {
#pragma clang fp contract(off)
      const size_t o = (((size_t)b * H + gh) * W + gw) * 4;
      f32x4 v = acc[0] + acc[1];
      v = v + bv;   // (the generic kernel's order: bias, residual, scale)
      if constexpr (SKIP) v = v + *reinterpret_cast<const f32x4*>(skip + o);
      v = v * p.scale;
      *reinterpret_cast<f32x4*>(out + o) = v;
    }
  }
}

template <bool ACT, bool SKIP>
int set_attr() {
  FD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_headf_kernel<ACT, SKIP>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(MAX_STEPS)));
  return FD_OK;
}

}  // namespace

bool fd_headf_supported(const ConvArgs& a, int ksize, int dtype) {
  if (!(dtype == FD_F32 && ksize == 3 && a.Cout == 4 && a.stats == nullptr)) return false;
  int chunks = 0;
  for (int s = 0; s < a.nseg; ++s) {
    if (a.seg[s].taps != 9 || a.seg[s].C % 4 != 0) return false;
    if ((a.seg[s].aff_off >= 0) != (a.affine != nullptr)) return false;
    chunks += (a.seg[s].C + CK - 1) / CK;
  }
  return chunks % 2 == 0 && chunks * 9 <= MAX_STEPS && (a.affine == nullptr || a.affC * 8 <= AFF_BYTES);
}

int fd_headf_init_attributes() {
  FD_TRY((set_attr<false, false>()));
  FD_TRY((set_attr<false, true>()));
  FD_TRY((set_attr<true, false>()));
  FD_TRY((set_attr<true, true>()));
  return FD_OK;
}

int fd_headf_launch(ConvArgs a, hipStream_t st) {
  a.tiles_h = fd_cdiv(a.H, TH);
  a.tiles_w = fd_cdiv(a.W, TW);
  a.tiles_n = 1;
  const long long nblk = (long long)a.B * a.tiles_h * a.tiles_w;
  FD_REQUIRE(nblk > 0 && nblk < (1ll << 31), "conv grid out of range");
  const dim3 grid((unsigned)nblk), block(NTH);
  int chunks = 0;
  for (int s = 0; s < a.nseg; ++s) chunks += (a.seg[s].C + CK - 1) / CK;
  const int lds = lds_bytes(chunks * 9);
  if (a.affine) {
    if (a.skip) hipLaunchKernelGGL((conv_headf_kernel<true, true>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((conv_headf_kernel<true, false>), grid, block, lds, st, a);
  } else {
    if (a.skip) hipLaunchKernelGGL((conv_headf_kernel<false, true>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((conv_headf_kernel<false, false>), grid, block, lds, st, a);
  }
  FD_LAUNCH_CHECK();
  return FD_OK;
}
