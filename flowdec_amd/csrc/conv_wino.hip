// conv_wino.hip -- 3x3 convolution as Winograd F(2,3) along W (time frames) x direct along H on the gfx950 matrix cores.
//
// Same contract as conv_mfma.hip (ddpm_conv3x3, flowdec/backbones/ncsnpp_utils/layers.py:128-134, with the ResnetBlockBigGANpp
// surroundings of layerspp.py:252-284 fused in: GroupNorm+SiLU operand transform, time/conv bias, folded 1x1 shortcut conv,
// residual, 1/sqrt(2), statistics of the output for the next GroupNorm, virtual channel concat) -- but 1.5x fewer MFMAs:
// for an output pair (y0, y1) = pixels (2j, 2j+1) of a row and the four inputs d0..d3 = pixels 2j-1 .. 2j+2
//   M0 = (d0 - d2) g0,   M1 = (d1 + d2) (g0+g1+g2)/2,   M2 = (d2 - d1) (g0-g1+g2)/2,   M3 = (d1 - d3) g2
//   y0 = M0 + M1 + M2,   y1 = M1 - M2 - M3                       (4 products instead of 6; g = one kernel row)
// summed over the three kernel rows dy and all input channels in the accumulators of position xi = 0..3.
//
// Why 1-D: the accumulators are the capacity limit.  F(2x2,3x3) needs 16 accumulator planes per output tile: with the 64 K
// accumulator registers of a workgroup the per-position GEMM tile shrinks to 64 x 64, whose weight stream (4096 / tiles
// bytes per MFMA cycle and CU) exceeds the 64 B/clk a CU gets from L2.  F(2,3) x direct needs 4 planes: 128 tiles (16 x 16
// pixels) x 128 couts per workgroup keeps the weight bytes per MFMA at the direct kernel's value.
//
// Mapping: one workgroup = 16 x 16 output pixels (128 Winograd tiles of 1 x 2) x 128 output channels, 8 waves; wave w owns
// position xi = w & 3 and the tile half ph = w >> 2 (two 32-tile patches of 4 rows x 8 tiles) x all 128 couts:
// 2 x 4 MFMA tiles of 32 x 32 = 128 accumulator registers, v_mfma_f32_32x32x16_f16.
//   * operands are fp16: the conv inputs are GroupNorm+SiLU outputs (bounded), fp16 keeps 3 more mantissa bits than bf16 and
//     -- unlike bf16 -- has packed add/fma, which the on-the-fly transform needs; raw (shortcut) inputs are clamped to the
//     fp16 range; storage stays bf16, accumulation f32;
//   * the 18 x 18 halo of a 32-channel chunk is activated (silu(a x + d)) and stored ONCE in LDS as fp16 (as the direct kernel
//     does in bf16); the Winograd input transform happens at fragment-read time: V_xi = z[col a(xi)] + s(xi) z[col b(xi)] is
//     two ds_read_b128 and four v_pk_fma_f16 per fragment, reused by the 4 cout tiles;
//   * weights are transformed at pack time: U_xi,dy = G(xi) . w[dy][0..2], stored [step = (chunk, dy)][xi][CoutPad][64 B]
//     with the direct kernel's XOR swizzle; a step's four 8 KiB slabs stream through a ring of 2 x 4 LDS slots by
//     global_load_lds; one barrier per step (16 MFMAs per wave);
//   * the folded 1x1 shortcut (Conv_2, layerspp.py:278-279) lands in the same accumulators: M0 += W x(y0), M3 -= W x(y1) for
//     the first 16 channels of a chunk and M1 += W/2 (x(y0) + x(y1)), M2 += W/2 (x(y0) - x(y1)) for the other 16, so all
//     four positions do half a K step each;
//   * epilogue: the four position planes of a tile meet in LDS (they live in four waves), y0 / y1 are formed there, then
//     bias / residual / scale / statistics / bf16 store exactly as in the direct kernel.
#include <type_traits>

#include "conv_common.h"

namespace {

using namespace fdconv;

typedef _Float16 f16;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

constexpr int cmax(int a, int b) { return a > b ? a : b; }

// ---- geometry -------------------------------------------------------------------------------------------------------------
constexpr int WP = 19;                 // halo row pitch in pixels: odd, so that consecutive rows differ by an odd number of
                                       // 16-byte slots (19 * 5 = 95) and a 2 rows x 8 tiles lane group covers all 16 slots
constexpr int NTH = 512, TH = 16, TW = 16, HH = TH + 2, HW = TW + 2, BN = 128;
constexpr int HALO_BYTES = HH * WP * ROWB;          // 27360
constexpr int SLAB = BN * WROWB;                    // 8 KiB: one (step, xi) weight slab
constexpr int NRING = 3;                            // ring depth in steps; LDS image [xi][slot][SLAB]: a wave's three slots are
                                                    // 8 KiB apart (ds_read immediates), step s lives in slot s % 3
constexpr int RING_BYTES = 4 * NRING * SLAB;        // 96 KiB
constexpr int HALO_OFF = RING_BYTES;
constexpr int AFF_OFF = HALO_OFF + 2 * HALO_BYTES;
constexpr int MAIN_BYTES = AFF_OFF + AFF_BYTES;     // 157120
constexpr int EP_ROWB = 64 * 4 + 16;                // staging row: 64 couts f32 + pad (17 slots: odd)
constexpr int EP_BYTES = 8 * 32 * EP_ROWB;          // [xi][ph][32 tiles] rows
constexpr int ST_REC = 144;                         // statistics record per thread: 32 floats + pad
constexpr int ST_BYTES = NTH * ST_REC;
constexpr int LDS_BYTES = cmax(MAIN_BYTES, cmax(2 * EP_BYTES, ST_BYTES));
constexpr int PPP = NTH / 4;                        // halo pixels per loader pass (4 slots per pixel)
constexpr int HITER = (HH * HW + PPP - 1) / PPP;    // 3
constexpr int CK = 32, EPS = 8;                     // channels per chunk / per 16-byte slot (bf16 storage, fp16 operands)
static_assert(HITER == 3, "halo conversion schedule assumes three slots per thread");
static_assert(HALO_BYTES % 16 == 0 && LDS_BYTES <= 160 * 1024, "LDS layout");

// MFMA column n (= lane & 31) -> (row, tile) inside a 4-row x 8-tile patch.  ds_read_b128 is served in the lane groups
// {0-3,12-15,20-27} {4-11,16-19,28-31} (+32): each group gets two whole rows, whose 16-byte slots are then all distinct.
__device__ __forceinline__ void tile_rc(int n, int& r, int& j) {
  if (n < 4) { r = 0; j = n; }
  else if (n < 12) { r = 2; j = n - 4; }
  else if (n < 16) { r = 0; j = n - 8; }
  else if (n < 20) { r = 3; j = n - 16; }
  else if (n < 28) { r = 1; j = n - 20; }
  else { r = 3; j = n - 24; }
}

__device__ __forceinline__ unsigned pack_f16(float a, float b) {
  f16x2 r = {(f16)a, (f16)b};
  return __builtin_bit_cast(unsigned, r);
}

// LDS-DMA of 64 x 16 B (one 1-KiB piece) issued from inline asm: hipcc does not count it, so every wait for it in this file
// is an explicit counted s_waitcnt vmcnt(N) (cdna_hip_programming.md 5.7: M0 written in the same statement that reads it).
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}


// ---- the MFMA cluster of one phase, hand-placed (inline asm): 8 MFMAs (2 patches x 4 cout tiles of one k-half) with the 8
// operand reads of the NEXT phase between them.  hipcc's scheduler otherwise pairs every MFMA with a wait for the read it
// has just issued; here nothing in the cluster waits, and the two waves of a SIMD take turns: one runs its cluster on the
// matrix pipe (s_setprio 1) while the other does its VALU segment (Winograd transform, halo conversion).
// The reads are invisible to hipcc's counters: wait_ops() is the matching s_waitcnt and (re)defines the registers for it.
__device__ __forceinline__ void wait_ops(u32x4 (&wf)[4], u32x4 (&ra)[2], u32x4 (&rb)[2]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(wf[3]), "+v"(ra[0]), "+v"(ra[1]), "+v"(rb[0]), "+v"(rb[1])
               :: "memory");
}
// wb / pa / pb: LDS byte addresses of the lane's first weight row / of its pixel in patch 0 for the two transform operands;
// WOFF, HOFF: compile-time offsets (ring slot; halo row, k-half).  Cout tiles are 2048 B apart (the swizzle does not depend on
// the tile), the second patch is 4 halo rows further.
template <int WOFF, int HOFF>
__device__ __forceinline__ void mma_cluster(f32x16 (&acc)[2][4], const u32x4 (&wf)[4], const u32x4 (&pf)[2], u32x4 (&nwf)[4],
                                            u32x4 (&nra)[2], u32x4 (&nrb)[2], int wb, int pa, int pb) {
  constexpr int P1 = 4 * WP * ROWB;
  asm volatile(
      "s_nop 1\n\t"
      "s_setprio 1\n\t"
      "v_mfma_f32_32x32x16_f16 %[a00], %[w0], %[p0], %[a00]\n\t"
      "ds_read_b128 %[nw0], %[wb] offset:%[o0]\n\t"
      "ds_read_b128 %[nw1], %[wb] offset:%[o1]\n\t"
      "v_mfma_f32_32x32x16_f16 %[a10], %[w0], %[p1], %[a10]\n\t"
      "ds_read_b128 %[nw2], %[wb] offset:%[o2]\n\t"
      "ds_read_b128 %[nw3], %[wb] offset:%[o3]\n\t"
      "v_mfma_f32_32x32x16_f16 %[a01], %[w1], %[p0], %[a01]\n\t"
      "ds_read_b128 %[na0], %[pa] offset:%[h0]\n\t"
      "ds_read_b128 %[nb0], %[pb] offset:%[h0]\n\t"
      "v_mfma_f32_32x32x16_f16 %[a11], %[w1], %[p1], %[a11]\n\t"
      "ds_read_b128 %[na1], %[pa] offset:%[h1]\n\t"
      "ds_read_b128 %[nb1], %[pb] offset:%[h1]\n\t"
      "v_mfma_f32_32x32x16_f16 %[a02], %[w2], %[p0], %[a02]\n\t"
      "v_mfma_f32_32x32x16_f16 %[a12], %[w2], %[p1], %[a12]\n\t"
      "v_mfma_f32_32x32x16_f16 %[a03], %[w3], %[p0], %[a03]\n\t"
      "v_mfma_f32_32x32x16_f16 %[a13], %[w3], %[p1], %[a13]\n\t"
      "s_setprio 0"
      : [a00] "+v"(acc[0][0]), [a10] "+v"(acc[1][0]), [a01] "+v"(acc[0][1]), [a11] "+v"(acc[1][1]), [a02] "+v"(acc[0][2]),
        [a12] "+v"(acc[1][2]), [a03] "+v"(acc[0][3]), [a13] "+v"(acc[1][3]), [nw0] "=&v"(nwf[0]), [nw1] "=&v"(nwf[1]),
        [nw2] "=&v"(nwf[2]), [nw3] "=&v"(nwf[3]), [na0] "=&v"(nra[0]), [na1] "=&v"(nra[1]), [nb0] "=&v"(nrb[0]), [nb1] "=&v"(nrb[1])
      : [w0] "v"(wf[0]), [w1] "v"(wf[1]), [w2] "v"(wf[2]), [w3] "v"(wf[3]), [p0] "v"(pf[0]), [p1] "v"(pf[1]), [wb] "v"(wb), [pa] "v"(pa),
        [pb] "v"(pb), [o0] "i"(WOFF), [o1] "i"(WOFF + 2048), [o2] "i"(WOFF + 4096), [o3] "i"(WOFF + 6144), [h0] "i"(HOFF), [h1] "i"(HOFF + P1)
      : "memory");
}
// the same reads without MFMAs (prologue)
template <int WOFF, int HOFF>
__device__ __forceinline__ void read_cluster(u32x4 (&nwf)[4], u32x4 (&nra)[2], u32x4 (&nrb)[2], int wb, int pa, int pb) {
  constexpr int P1 = 4 * WP * ROWB;
  asm volatile(
      "ds_read_b128 %[nw0], %[wb] offset:%[o0]\n\t"
      "ds_read_b128 %[nw1], %[wb] offset:%[o1]\n\t"
      "ds_read_b128 %[nw2], %[wb] offset:%[o2]\n\t"
      "ds_read_b128 %[nw3], %[wb] offset:%[o3]\n\t"
      "ds_read_b128 %[na0], %[pa] offset:%[h0]\n\t"
      "ds_read_b128 %[nb0], %[pb] offset:%[h0]\n\t"
      "ds_read_b128 %[na1], %[pa] offset:%[h1]\n\t"
      "ds_read_b128 %[nb1], %[pb] offset:%[h1]"
      : [nw0] "=&v"(nwf[0]), [nw1] "=&v"(nwf[1]), [nw2] "=&v"(nwf[2]), [nw3] "=&v"(nwf[3]), [na0] "=&v"(nra[0]), [na1] "=&v"(nra[1]),
        [nb0] "=&v"(nrb[0]), [nb1] "=&v"(nrb[1])
      : [wb] "v"(wb), [pa] "v"(pa), [pb] "v"(pb), [o0] "i"(WOFF), [o1] "i"(WOFF + 2048), [o2] "i"(WOFF + 4096), [o3] "i"(WOFF + 6144),
        [h0] "i"(HOFF), [h1] "i"(HOFF + P1)
      : "memory");
}

template <bool ACT, bool SKIP>
__global__ __launch_bounds__(NTH, 2) void conv_wino_kernel(ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const hbuf = smem + HALO_OFF;
  char* const afftab = smem + AFF_OFF;
  FD_T2(const unsigned long long t2_entry = __builtin_amdgcn_s_memtime();)

  // ---- tile decode with XCD-aware remap (as conv_mfma.hip) ------------------------------------------------------------
  const int bid = blockIdx.x, nblk = gridDim.x;
  int lid;
  {
    const int xcd = bid & 7, qq = nblk >> 3, rr = nblk & 7;
    lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  }
  const int nt_i = lid % p.tiles_n;
  int pt = lid / p.tiles_n;
  const int tw_i = pt % p.tiles_w; pt /= p.tiles_w;
  const int th_i = pt % p.tiles_h;
  const int b = pt / p.tiles_h;
  const int h0 = th_i * TH, w0 = tw_i * TW, n0 = nt_i * BN;
  const int H = p.H, W = p.W;

  const int t = threadIdx.x;
  const int q = t & 3;        // 16-byte slot inside the 64-byte chunk row
  const int prow = t >> 2;

  // ---- halo loader.  Loads are unconditional (clamped addresses); validity is applied when the converted registers are
  // written to LDS.  Threads whose third slot lies past the 18 x 18 halo redo their second one (same load, same value, same
  // LDS address) so that the conversion code has no divergent branch and can be interleaved with the MFMAs. -----------------
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
    hlds[i] = (hr * WP + hc) * ROWB + q * 16;
    if (ok) pvalid |= 1u << i;
  }
  const size_t img_elems = (size_t)H * W;
  u32x4 hreg[HITER];
  __amdgpu_buffer_rsrc_t nsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, 16, 0x00020000);
  int nC = 0, nc = 0, naff = 0;
  unsigned vmask[HITER] = {0u, 0u, 0u};   // all ones where the slot holds real data (pixel inside the image, channel < C)
  // cursor of the chunk whose halo is loaded next (runs ahead of the chunk being multiplied)
  int cs = 0, cch = -1;
  bool cur_end = false;
  auto next_chunk = [&]() {   // advance the cursor and set up the load / conversion state of that chunk
    if (!cur_end) {
      ++cch;
      if (cch >= (p.seg[cs].C + CK - 1) / CK) { ++cs; cch = 0; }
      if (cs >= p.nseg) cur_end = true;
    }
    if (cur_end) return;
    const Seg sg = p.seg[cs];
    const bf16* src = reinterpret_cast<const bf16*>(sg.src) + (size_t)b * img_elems * sg.C;
    nsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(src), 0, (int)(img_elems * sg.C * sizeof(bf16)), 0x00020000);
    nC = sg.C;
    const int c = cch * CK + q * EPS;
    const bool nchan_ok = c < sg.C;
    nc = nchan_ok ? c : 0;
    naff = sg.aff_off >= 0 ? (sg.aff_off + nc) * 8 : 0;
#pragma unroll
    for (int i = 0; i < HITER; ++i) vmask[i] = (nchan_ok && ((pvalid >> i) & 1u)) ? 0xffffffffu : 0u;
  };
  // past the end of the K loop every lane re-reads element 0 of the last tensor (one cache line; the data is never used)
  auto load_halo = [&]() {
    const int on = cur_end ? 0 : 1;
#pragma unroll
    for (int i = 0; i < HITER; ++i) hreg[i] = __builtin_amdgcn_raw_buffer_load_b128(nsrd, (pixl[i] * nC + nc) * 2 * on, 0, 0);
  };
  // conversion of one 32-bit word (two channels) of a loaded slot, in place: bf16 -> [silu(a x + d)] -> fp16, zero padding
  auto conv_word = [&](auto act_tag, int i, int j) {
    constexpr bool A = decltype(act_tag)::value;
    const unsigned u = hreg[i][j];
    const float x0 = __builtin_bit_cast(float, u << 16), x1 = __builtin_bit_cast(float, u & 0xffff0000u);
    unsigned r;
    if constexpr (A) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(afftab + naff + 16 * j);
      r = pack_f16(fd_silu(fmaf(x0, a[0], a[1])), fd_silu(fmaf(x1, a[2], a[3])));
    } else {   // raw input (activated / resampled upstream, or the shortcut input): saturate at HALF the fp16 range, so that the
               // transform z[a] +- z[b] (packed fp16) cannot overflow to inf either.  Dynamic-range contract: MEASUREMENTS.md "Parity detail".
      r = pack_f16(__builtin_amdgcn_fmed3f(x0, -32752.f, 32752.f), __builtin_amdgcn_fmed3f(x1, -32752.f, 32752.f));
    }
    // zero padding AFTER the activation -- as an AND: a select around the SiLU becomes a divergent branch per word, and a branch
    // ends the basic block the MFMA interleave works in
    hreg[i][j] = r & vmask[i];
  };
  auto store_slot = [&](int i, int buf) { *reinterpret_cast<u32x4*>(hbuf + buf * HALO_BYTES + hlds[i]) = hreg[i]; };
  // words [w0, w1) of the 12 words of a chunk's three slots; a slot is stored as soon as its fourth word is converted
  auto conv_words = [&](auto act_tag, int wa, int wb, int buf) {
#pragma unroll
    for (int w = 0; w < 12; ++w) {
      if (w < wa || w >= wb) continue;
      conv_word(act_tag, w >> 2, w & 3);
      if ((w & 3) == 3) store_slot(w >> 2, buf);
    }
  };

  // ---- weight ring: the four slabs [xi][128 rows][64 B] of step `stp` -> slot stp % 3; 32 pieces of 1 KiB, 4 per wave
  // (piece k of a wave = its 1-KiB slice of slab xi = k) ------------------------------------------------------------------
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const char* const wsrc = reinterpret_cast<const char*>(p.w) + (size_t)n0 * WROWB + wave * 1024 + lane * 16;
  const size_t xs = (size_t)p.CoutPad * WROWB;   // bytes between the slabs of consecutive positions
  auto dma_step = [&](int stp, int slot) {
    const char* src = wsrc + (size_t)stp * 4 * xs;
    const unsigned dst = (unsigned)(slot * SLAB + wave * 1024);
#pragma unroll
    for (int k = 0; k < 4; ++k) glds16(src + k * xs, dst + k * NRING * SLAB);
  };

  // ---- per-lane fragment coordinates --------------------------------------------------------------------------------------
  const int xi = wave & 3, ph = wave >> 2;
  const int l31 = lane & 31, lh = lane >> 5;
  int pr, pj;
  tile_rc(l31, pr, pj);
  // V_xi = z[col al] + sg * z[col be]  (halo columns relative to the tile's first input column 2j)
  int al = xi == 0 ? 0 : (xi == 2 ? 2 : 1);
  int be = xi == 0 ? 2 : (xi == 1 ? 2 : (xi == 2 ? 1 : 3));
  float sg = xi == 1 ? 1.f : -1.f;
  int pa0, pb0;   // patch 0 of this wave in halo buffer 0 (patch 1 = 4 halo rows further)
  auto set_pixel_bases = [&](int koff) {
    const int base = HALO_OFF + ((4 * (2 * ph) + pr) * WP + 2 * pj) * ROWB + lh * 16 + koff;
    pa0 = base + al * ROWB;
    pb0 = base + be * ROWB;
  };
  set_pixel_bases(0);
  f16x2 sig2 = {(f16)sg, (f16)sg};
  // [k-half]: slabs of xi, row = cout (tile nj = + nj * 2048), 16-B column (2 * ks + lh) ^ swizzle(row)
  const int wsw = (l31 >> 2) & 3;
  int wb0 = xi * NRING * SLAB + l31 * WROWB + ((lh ^ wsw) * 16);
  const int wb1 = xi * NRING * SLAB + l31 * WROWB + (((2 + lh) ^ wsw) * 16);

  f32x16 acc[2][4];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int nj = 0; nj < 4; ++nj)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][nj][e] = 0.f;

  int n3 = 0, n1 = 0;
  for (int s = 0; s < p.nseg; ++s) {
    const int nch = (p.seg[s].C + CK - 1) / CK;
    if (p.seg[s].taps == 9) n3 += nch; else n1 += nch;
  }
  const int nsteps = n3 * 3 + n1;   // weight steps: (chunk, dy) for the 3x3 part, one per shortcut chunk
  const int last_step = nsteps - 1;

  // Workgroup barrier of the K loop.  Everything this wave has in flight on the vector-memory counter is, oldest first:
  // ..., the weight pieces of step s+1 (must have landed: they are read right after this barrier), the 4 pieces issued after
  // the previous barrier, and possibly 3 halo loads issued after those.  NKEEP = how many of the youngest may stay in flight.
  auto step_barrier = [&](auto keep_tag) {
    constexpr int NKEEP = decltype(keep_tag)::value;
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NKEEP == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if constexpr (NKEEP == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  auto lds_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  using K7 = std::integral_constant<int, 7>;
  using K4 = std::integral_constant<int, 4>;
  using K0 = std::integral_constant<int, 0>;

  u32x4 wfA[4], wfB[4], raA[2], rbA[2], raB[2], rbB[2], pfA[2], pfB[2];
  // Winograd input transform of a fragment: 4 x v_pk_fma_f16
  auto combine = [&](u32x4 (&pf)[2], const u32x4 (&ra)[2], const u32x4 (&rb)[2]) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned ua = ra[mi][j], ub = rb[mi][j];   // (element -> scalar first: see conv_mfma.hip on bit_cast of vector elements)
        const f16x2 va = __builtin_bit_cast(f16x2, ua), vb = __builtin_bit_cast(f16x2, ub);
        const f16x2 r = __builtin_elementwise_fma(vb, sig2, va);
        pf[mi][j] = __builtin_bit_cast(unsigned, r);
      }
  };
  // the shortcut steps use other operand selectors: positions 0 / 3 take k-half 0 with x(y0) / x(y1) (weights W / -W),
  // positions 1 / 2 take k-half 1 with x(y0) +- x(y1) (weights W / 2)
  auto switch_to_shortcut = [&]() {
    al = xi == 3 ? 2 : 1;
    be = xi == 0 ? 1 : 2;
    sg = xi == 1 ? 1.f : (xi == 2 ? -1.f : 0.f);
    const int ksc = (xi == 1 || xi == 2) ? 1 : 0;
    set_pixel_bases(32 * ksc);
    sig2 = f16x2{(f16)sg, (f16)sg};
    wb0 = xi * NRING * SLAB + l31 * WROWB + (((2 * ksc + lh) ^ wsw) * 16);
  };
  using TACT = std::integral_constant<bool, ACT>;
  using TRAW = std::integral_constant<bool, false>;

  // ---- prologue: halo of chunk 0, the first three weight steps and the affine table in one memory round trip; then the halo
  // of chunk 1 goes into the registers (it is converted during chunk 0) ---------------------------------------------------
  next_chunk();
  load_halo();
  dma_step(0, 0);
  dma_step(last_step >= 1 ? 1 : last_step, 1);
  dma_step(last_step >= 2 ? 2 : last_step, 2);
  if (ACT) {
    const float* ap = p.affine + (size_t)b * p.affC * 2;
    for (int i = t; i < p.affC / 2; i += NTH) *reinterpret_cast<f32x4*>(afftab + i * 16) = *reinterpret_cast<const f32x4*>(ap + i * 4);
    __syncthreads();
  }
  conv_words(TACT{}, 0, 12, 0);
  __builtin_amdgcn_sched_barrier(0);
  next_chunk();
  load_halo();
  step_barrier(K7{});   // steps 0 and 1 landed (step 2 and the halo loads may still fly); halo 0 published
  FD_T2(const unsigned long long t2_first = __builtin_amdgcn_s_memtime();)
  read_cluster<0, 0>(wfA, raA, rbA, wb0, pa0, pb0);

  int step = 0, hcur = 0;
  constexpr int CENTER = WP * ROWB;   // row offset of the output row itself (dy = 1)

  // ---- 3x3 part.  Per chunk three steps (dy), each 2 k-halves x 8 MFMAs with ONE barrier in the middle (operand fragments
  // are software-pipelined across it).  Step (chunk, dy) reads ring slot dy; after its barrier the slot is refilled with step
  // s+3, which has two full steps to land.  The halo of chunk c+1 sits in registers since the end of chunk c-1 (loaded at its
  // last phase) and is converted + stored during the phases B0 A1 B1 A2 of chunk c, three words each; the loads for chunk c+2
  // follow in phase B2.  sched_barrier(0) pins the phase boundaries (hipcc otherwise hoists the conversion -- and its wait --
  // right behind the loads).
  auto chunk_body = [&](auto next_tag, bool to_shortcut) {
    const int nb = hcur ^ 1;
    const int pac = pa0 + hcur * HALO_BYTES, pbc = pb0 + hcur * HALO_BYTES;
    // ---------------- dy = 0
    wait_ops(wfA, raA, rbA);
    combine(pfA, raA, rbA);
    mma_cluster<0 * SLAB, 0 * CENTER + 32>(acc, wfA, pfA, wfB, raB, rbB, wb1, pac, pbc);
    step_barrier(K7{});
    wait_ops(wfB, raB, rbB);
    combine(pfB, raB, rbB);
    conv_words(next_tag, 0, 3, nb);
    mma_cluster<1 * SLAB, 1 * CENTER>(acc, wfB, pfB, wfA, raA, rbA, wb0, pac, pbc);
    // (the weight pieces go out at the END of the phase: hipcc does not count them, so its own wait for the first halo word
    // above -- vmcnt(2) by its count -- would otherwise also cover pieces issued in front of it)
    dma_step(step + 3 <= last_step ? step + 3 : last_step, 0);
    // ---------------- dy = 1
    wait_ops(wfA, raA, rbA);
    combine(pfA, raA, rbA);
    conv_words(next_tag, 3, 6, nb);
    mma_cluster<1 * SLAB, 1 * CENTER + 32>(acc, wfA, pfA, wfB, raB, rbB, wb1, pac, pbc);
    step_barrier(K4{});
    wait_ops(wfB, raB, rbB);
    combine(pfB, raB, rbB);
    conv_words(next_tag, 6, 9, nb);
    mma_cluster<2 * SLAB, 2 * CENTER>(acc, wfB, pfB, wfA, raA, rbA, wb0, pac, pbc);
    dma_step(step + 4 <= last_step ? step + 4 : last_step, 1);
    // ---------------- dy = 2
    wait_ops(wfA, raA, rbA);
    combine(pfA, raA, rbA);
    conv_words(next_tag, 9, 12, nb);
    mma_cluster<2 * SLAB, 2 * CENTER + 32>(acc, wfA, pfA, wfB, raB, rbB, wb1, pac, pbc);
    step_barrier(K4{});   // halo of the next chunk published
    wait_ops(wfB, raB, rbB);
    combine(pfB, raB, rbB);
    next_chunk();
    load_halo();
    if (to_shortcut) {
      switch_to_shortcut();
      mma_cluster<0 * SLAB, CENTER>(acc, wfB, pfB, wfA, raA, rbA, wb0, pa0 + nb * HALO_BYTES, pb0 + nb * HALO_BYTES);
    } else {
      mma_cluster<0 * SLAB, 0>(acc, wfB, pfB, wfA, raA, rbA, wb0, pa0 + nb * HALO_BYTES, pb0 + nb * HALO_BYTES);
    }
    dma_step(step + 5 <= last_step ? step + 5 : last_step, 2);
    step += 3;
    hcur ^= 1;
  };
  // (sequential loops, each with ONE set of MFMA sites: with both bodies inside one loop hipcc keeps two copies of the
  // accumulators -- see conv_mfma.hip)
  const int nmain = n1 > 0 ? n3 - 1 : n3;
  for (int i = 0; i < nmain; ++i) chunk_body(TACT{}, false);
  for (int i = nmain; i < n3; ++i) chunk_body(TRAW{}, true);   // last 3x3 chunk in front of the shortcut: the halo it converts is raw
  FD_T2(const unsigned long long t2_sc = __builtin_amdgcn_s_memtime();)
  // ---- folded 1x1 shortcut: one step per chunk, half a K step (8 MFMAs) per wave.  Same pipeline with short steps: the halo
  // of the next chunk is in registers, converted and stored before the barrier; the loads of the chunk after it follow.
  for (int i = 0; i < n1; ++i) {
    const int nb = hcur ^ 1;
    const int slot = step % NRING, slotn = (step + 1) % NRING;
    wait_ops(wfA, raA, rbA);
    combine(pfA, raA, rbA);
    conv_words(TRAW{}, 0, 12, nb);
    __builtin_amdgcn_sched_barrier(0);
    next_chunk();
    load_halo();
    // 8 MFMAs of this step; the operand reads of the next step follow the barrier (they need its halo and weights)
    mma_cluster<0, 0>(acc, wfA, pfA, wfB, raB, rbB, wb0, pa0, pb0);   // (the reads of this call are dummies)
    step_barrier(K7{});
    dma_step(step + 3 <= last_step ? step + 3 : last_step, slot);
    wait_ops(wfB, raB, rbB);
    read_cluster<0, CENTER>(wfA, raA, rbA, wb0 + slotn * SLAB, pa0 + nb * HALO_BYTES, pb0 + nb * HALO_BYTES);
    ++step; hcur ^= 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // all fragment reads / DMA done before the epilogue reuses the LDS
  FD_T2(const unsigned long long t2_loop = __builtin_amdgcn_s_memtime();)

  // ---- epilogue -------------------------------------------------------------------------------------------------------
  // 4 rounds (mi, nh): every wave stages acc[mi][2nh..2nh+1] (32 tiles x 64 couts of its position) as
  // [xi][ph][tile][cout] f32; then thread (tile, 8 couts) reads the four positions of its tile, forms y0 / y1 and finishes.
  bf16* out = reinterpret_cast<bf16*>(p.out);
  const bf16* skip = SKIP ? reinterpret_cast<const bf16*>(p.skip) : nullptr;
  const int oct = t & 7, tl = t >> 3;
  const int ph_t = tl >> 5, n_t = tl & 31;
  int pr_t, pj_t;
  tile_rc(n_t, pr_t, pj_t);
  float bv[2][8];
#pragma unroll
  for (int nh = 0; nh < 2; ++nh) {
    const int n_e = n0 + nh * 64 + oct * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) bv[nh][j] = 0.f;
    if (p.bias && n_e < p.Cout) {
      const float* bp = p.bias + (size_t)(p.bias_rows > 1 ? b : 0) * p.Cout + n_e;
#pragma unroll
      for (int j = 0; j < 8; ++j) bv[nh][j] = bp[j];
    }
  }
  // inverse of the per-cout power-of-two scale of the packed weights (wino_scale_kernel): applied exactly, inside the fma that adds the
  // residual / the bias
  float iv[2][8];
#pragma unroll
  for (int nh = 0; nh < 2; ++nh) {
    const float* ip = p.w_scale + n0 + nh * 64 + oct * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) iv[nh][j] = ip[j];
  }
  float ssum[2][8], ssq[2][8];
#pragma unroll
  for (int nh = 0; nh < 2; ++nh)
#pragma unroll
    for (int j = 0; j < 8; ++j) ssum[nh][j] = ssq[nh][j] = 0.f;

#pragma unroll
  for (int rd = 0; rd < 4; ++rd) {
    const int mi = rd >> 1, nh = rd & 1;
    char* const stage = smem + (rd & 1) * EP_BYTES;
    const int n_e = n0 + nh * 64 + oct * 8;
    const bool n_ok = n_e < p.Cout;
    const int gh = h0 + 4 * (2 * ph_t + mi) + pr_t, gw = w0 + 2 * pj_t;
    const bool v0 = n_ok && gh < H && gw < W, v1 = n_ok && gh < H && gw + 1 < W;
    const size_t oaddr = v0 ? (((size_t)b * H + gh) * W + gw) * p.Cout + n_e : (size_t)0;
    const size_t oaddr1 = v1 ? oaddr + p.Cout : (size_t)0;
    u32x4 sk0, sk1;
    if constexpr (SKIP) {   // residual prefetch: latency hides behind the staging round trip
      sk0 = *reinterpret_cast<const u32x4*>(skip + oaddr);
      sk1 = *reinterpret_cast<const u32x4*>(skip + oaddr1);
    }
    {
      char* dst = stage + ((xi * 2 + ph) * 32 + l31) * EP_ROWB + (4 * lh) * 4;
#pragma unroll
      for (int qq = 0; qq < 2; ++qq)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const f32x16& a = acc[mi][2 * nh + qq];
          f32x4 v = {a[4 * qd], a[4 * qd + 1], a[4 * qd + 2], a[4 * qd + 3]};
          *reinterpret_cast<f32x4*>(dst + (qq * 32 + 8 * qd) * 4) = v;
        }
    }
    lds_barrier();
    float y0[8], y1[8];
    {
      const char* src = stage + (ph_t * 32 + n_t) * EP_ROWB + oct * 32;
      f32x4 m[4][2];
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        m[x][0] = *reinterpret_cast<const f32x4*>(src + x * 64 * EP_ROWB);
        m[x][1] = *reinterpret_cast<const f32x4*>(src + x * 64 * EP_ROWB + 16);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float m0 = m[0][j >> 2][j & 3], m1 = m[1][j >> 2][j & 3], m2 = m[2][j >> 2][j & 3], m3 = m[3][j >> 2][j & 3];
        y0[j] = m0 + m1 + m2;
        y1[j] = m1 - m2 - m3;
      }
    }
    if constexpr (SKIP) {
      const bf16x8 s0 = __builtin_bit_cast(bf16x8, sk0), s1 = __builtin_bit_cast(bf16x8, sk1);
#pragma unroll
      for (int j = 0; j < 8; ++j) { y0[j] = fmaf(y0[j], iv[nh][j], (float)s0[j]); y1[j] = fmaf(y1[j], iv[nh][j], (float)s1[j]); }
    }
    const float k0 = v0 ? 1.f : 0.f, k1 = v1 ? 1.f : 0.f;   // pixels outside the image do not enter the statistics
    bf16x8 t0, t1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if constexpr (SKIP) {
        y0[j] = (y0[j] + bv[nh][j]) * p.scale;
        y1[j] = (y1[j] + bv[nh][j]) * p.scale;
      } else {
        y0[j] = fmaf(y0[j], iv[nh][j], bv[nh][j]) * p.scale;
        y1[j] = fmaf(y1[j], iv[nh][j], bv[nh][j]) * p.scale;
      }
      const float a0 = y0[j] * k0, a1 = y1[j] * k1;
      ssum[nh][j] += a0 + a1;
      ssq[nh][j] = fmaf(a0, a0, fmaf(a1, a1, ssq[nh][j]));
      t0[j] = (bf16)y0[j]; t1[j] = (bf16)y1[j];
    }
    if (v0) *reinterpret_cast<u32x4*>(out + oaddr) = __builtin_bit_cast(u32x4, t0);
    if (v1) *reinterpret_cast<u32x4*>(out + oaddr1) = __builtin_bit_cast(u32x4, t1);
  }
  lds_barrier();

  if (p.stats) {   // per-tile partial sums of the output: reduce over the 64 threads that share a cout octet
    char* rec = smem + t * ST_REC;
#pragma unroll
    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<f32x4*>(rec + nh * 64 + 16 * j) = f32x4{ssum[nh][2 * j], ssq[nh][2 * j], ssum[nh][2 * j + 1], ssq[nh][2 * j + 1]};
    lds_barrier();
    const int tile = th_i * p.tiles_w + tw_i;
    if (t < 2 * BN) {   // o = 2 * channel + which; channel = nh * 64 + oct * 8 + j
      const int c = t >> 1, which = t & 1;
      const int nh = c >> 6, oc = (c >> 3) & 7, j = c & 7;
      const char* src = smem + oc * ST_REC + nh * 64 + (2 * j + which) * 4;
      float a = 0.f;
#pragma unroll 8
      for (int r = 0; r < 64; ++r) a += *reinterpret_cast<const float*>(src + r * 8 * ST_REC);
      if (n0 + c < p.CoutPad) p.stats[(((size_t)b * p.tiles_h * p.tiles_w + tile) * p.CoutPad + n0) * 2 + t] = a;
    }
  }
  FD_T2(
  if (p.dbg && t == 0 && bid < 8192) {
    const unsigned long long t2_end = __builtin_amdgcn_s_memtime();
    unsigned long long* d = p.dbg + (size_t)bid * 8;
    d[0] = t2_first - t2_entry; d[1] = t2_loop - t2_first; d[2] = t2_end - t2_loop; d[3] = t2_sc - t2_first; d[4] = t2_loop - t2_sc;
    d[5] = 0; d[6] = 0;
  }
  )
}

// ---- weight packing: [Cout][Cin][3][3] f32 -> [step = (segment, chunk, dy)][xi][CoutPad][64 B] fp16 ---------------------
// U_xi = G(xi) . w[dy][0..2] with G = {(1,0,0), (.5,.5,.5), (.5,-.5,.5), (0,0,1)}; shortcut chunks (taps == 1):
// one step per chunk with the slabs {W, W/2, W/2, -W} (see the kernel header).  16-byte columns XOR-swizzled by (cout >> 2) & 3.
// Per-cout power-of-two scale (fp16 goes subnormal below 6.1e-5; a checkpoint may hold tiny weights in the layers the reference
// zero-initialises, layers.py:100): one block per cout, m = max |U| over the 3x3 rows and the folded shortcut; tab[n] = 2^-k, tab[CoutPad +
// n] = 2^k with m 2^k in [2^(U_EXP-1), 2^U_EXP).  Both weight sets of a cout get the same factor: they share the accumulators.
constexpr int U_EXP = 9;
__global__ void wino_scale_kernel(const float* __restrict__ w, const float* __restrict__ w_sc, float* __restrict__ tab, int Cout, int CoutPad,
                                  int Cin, int S) {
  __shared__ float red[256];
  const int n = blockIdx.x, t = threadIdx.x;
  float m = 0.f;
  if (n < Cout) {
    for (int i = t; i < Cin * 3; i += 256) {
      const float* g = w + ((size_t)n * Cin * 3 + i) * 3;
      m = fmaxf(m, fmaxf(fmaxf(fabsf(g[0]), fabsf(g[2])), fmaxf(fabsf(0.5f * (g[0] + g[1] + g[2])), fabsf(0.5f * (g[0] - g[1] + g[2])))));
    }
    if (w_sc)
      for (int i = t; i < S; i += 256) m = fmaxf(m, fabsf(w_sc[(size_t)n * S + i]));
  }
  red[t] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) red[t] = fmaxf(red[t], red[t + o]);
    __syncthreads();
  }
  if (t == 0) {
    int k = 0;
    m = red[0];
    if (m > 0.f && m < 3.0e38f) {
      int e;
      frexpf(m, &e);
      k = U_EXP - e;
      k = k < -100 ? -100 : (k > 100 ? 100 : k);
    }
    tab[n] = ldexpf(1.f, -k);
    tab[CoutPad + n] = ldexpf(1.f, k);
  }
}

__global__ void wino_pack_kernel(const float* __restrict__ w, char* __restrict__ dst, const float* __restrict__ tab, int Cout, int CoutPad,
                                 int C0, int C1, int taps, long long step0) {
  const int nchunk0 = (C0 + CK - 1) / CK, nchunks = nchunk0 + (C1 + CK - 1) / CK;
  const int spc = taps == 9 ? 3 : 1;   // steps per chunk
  const long long total = (long long)nchunks * spc * 4 * CoutPad * CK;
  const int Cin = C0 + C1;
  f16* d = reinterpret_cast<f16*>(dst + step0 * 4 * CoutPad * WROWB);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i % CK);
    long long r = i / CK;
    const int n = (int)(r % CoutPad); r /= CoutPad;
    const int x = (int)(r % 4); r /= 4;
    const int dy = (int)(r % spc);
    const int chunk = (int)(r / spc);
    float v = 0.f;
    if (n < Cout) {
      int c;
      if (chunk < nchunk0) { c = chunk * CK + k; if (c >= C0) c = -1; }
      else { c = (chunk - nchunk0) * CK + k; c = (c < C1) ? C0 + c : -1; }
      if (c >= 0) {
        if (taps == 9) {
          const float* g = w + (((size_t)n * Cin + c) * 3 + dy) * 3;
          v = x == 0 ? g[0] : (x == 1 ? 0.5f * (g[0] + g[1] + g[2]) : (x == 2 ? 0.5f * (g[0] - g[1] + g[2]) : g[2]));
        } else {
          const float g = w[(size_t)n * Cin + c];
          v = x == 0 ? g : (x == 3 ? -g : 0.5f * g);
        }
      }
    }
    const int col = (k / 8) ^ ((n >> 2) & 3);
    d[(i - k) + col * 8 + (k % 8)] = (f16)(v * tab[CoutPad + n]);
  }
}

inline int pad_to(int x, int a) { return (x + a - 1) / a * a; }
inline long long wino_steps(int C0, int C1, int spc) { return (long long)(fd_cdiv(C0, CK) + fd_cdiv(C1, CK)) * spc; }

}  // namespace

bool fd_wino_supported(int Cout, int C0, int C1, int S0, int S1, int ksize) {
  // Cout: multiples of 128 whose padded width equals the direct kernel's statistics stride (fd_conv_cout_pad: 128, then multiples of
  // 256) -- the per-tile GroupNorm partials of both kernels share one layout [B][tiles][fd_conv_cout_pad(Cout)][2]
  return ksize == 3 && Cout > 0 && Cout % BN == 0 && pad_to(Cout, BN) == fd_conv_cout_pad(Cout) && C0 > 0 && C0 % CK == 0 && C1 % CK == 0 &&
         S0 % CK == 0 && S1 % CK == 0;
}

namespace {
// The packed buffer starts with a header that holds the scale table [2][CoutPad] f32 (inverse scales, scales): a position that
// depends on the weight alone, not on the segment list of a launch (conv_wino4.hip does the same); the packed steps follow.
long long wino_hdr_bytes(int CoutPad) { return (2ll * CoutPad * (long long)sizeof(float) + 4095) / 4096 * 4096; }
}  // namespace

long long fd_wino_packed_bytes(int Cout, int C0, int C1, int S0, int S1) {
  const int CoutPad = pad_to(Cout, BN);
  return wino_hdr_bytes(CoutPad) + (wino_steps(C0, C1, 3) + wino_steps(S0, S1, 1)) * 4 * CoutPad * WROWB + 1024;
}

int fd_wino_pack_weights(const float* w, const float* w_sc, void* packed, int Cout, int C0, int C1, int S0, int S1, hipStream_t st) {
  const int CoutPad = pad_to(Cout, BN);
  float* const tab = reinterpret_cast<float*>(packed);
  char* const steps = reinterpret_cast<char*>(packed) + wino_hdr_bytes(CoutPad);
  hipLaunchKernelGGL(wino_scale_kernel, dim3(CoutPad), dim3(256), 0, st, w, w_sc, tab, Cout, CoutPad, C0 + C1, w_sc ? S0 + S1 : 0);
  auto run = [&](const float* src, int c0, int c1, int taps, long long step0) {
    const long long total = wino_steps(c0, c1, taps == 9 ? 3 : 1) * 4 * CoutPad * CK;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(wino_pack_kernel, dim3(blocks), dim3(256), 0, st, src, steps, tab, Cout, CoutPad, c0, c1, taps, step0);
  };
  run(w, C0, C1, 9, 0);
  if (w_sc) run(w_sc, S0, S1, 1, wino_steps(C0, C1, 3));
  FD_LAUNCH_CHECK();
  return FD_OK;
}

int fd_wino_init_attributes() {
  FD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  FD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  FD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  FD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  return FD_OK;
}

int fd_wino_launch(ConvArgs a, hipStream_t st) {
  a.tiles_h = fd_cdiv(a.H, TH);
  a.tiles_w = fd_cdiv(a.W, TW);
  a.tiles_n = a.Cout / BN;
  a.CoutPad = pad_to(a.Cout, BN);
  a.w_scale = reinterpret_cast<const float*>(a.w);                                // header of the packed buffer
  a.w = reinterpret_cast<const char*>(a.w) + wino_hdr_bytes(a.CoutPad);           // the packed steps
  const long long nblk = (long long)a.B * a.tiles_h * a.tiles_w * a.tiles_n;
  FD_REQUIRE(nblk > 0 && nblk < (1ll << 31), "conv grid out of range");
  const dim3 grid((unsigned)nblk), block(NTH);
  if (a.affine) {
    if (a.skip) hipLaunchKernelGGL((conv_wino_kernel<true, true>), grid, block, LDS_BYTES, st, a);
    else hipLaunchKernelGGL((conv_wino_kernel<true, false>), grid, block, LDS_BYTES, st, a);
  } else {
    if (a.skip) hipLaunchKernelGGL((conv_wino_kernel<false, true>), grid, block, LDS_BYTES, st, a);
    else hipLaunchKernelGGL((conv_wino_kernel<false, false>), grid, block, LDS_BYTES, st, a);
  }
  FD_LAUNCH_CHECK();
  return FD_OK;
}
