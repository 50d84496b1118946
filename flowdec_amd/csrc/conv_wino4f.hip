// conv_wino4f.hip -- conv_wino4.hip (Winograd F(4,3) along W x direct along H, 256-cout workgroups) in EXACT FLOAT32: f32 storage, f32
// transforms, v_mfma_f32_32x32x2_f32 (bit-exact f32 FMA chains), f32 accumulation.  The fp32 mode of the model (BASELINE config 5's dtype;
// the reference's inference arithmetic: no autocast anywhere) ran on the direct kernel only, at 0.87 of the 157 TFLOP/s
// f32 matrix peak -- a kernel that cannot get faster.  F(4,3) executes HALF the MFMAs (6 products per 4 outputs and kernel row instead
// of 12), and with the f32 matrix instruction 16 x slower than the fp16 one everything that holds the fp16 kernel back (weight stream,
// producer, epilogue: conv_wino4.hip, MEASUREMENTS.md R4-R6) is an order of magnitude below the MFMA time here.
//
// Same contract and same dataflow as conv_wino4.hip (ddpm_conv3x3, flowdec/backbones/ncsnpp_utils/layers.py:128-134, with the
// ResnetBlockBigGANpp surroundings of layerspp.py:252-284 fused in); what differs:
//   * a K chunk is 8 channels (one 32-byte LDS row = two 16-byte halves of 4 floats) instead of 16: every LDS byte count, every ring slot
//     and every DMA piece of the fp16 kernel keeps its size; the halo of a chunk pair is still 64 contiguous bytes per pixel;
//   * a fragment quad (16 bytes per lane) holds 4 floats: lane (l31, lh) supplies channel 4 lh + j of row l31 to the j-th of FOUR
//     v_mfma_f32_32x32x2_f32 per fragment pair (the matrix instruction contracts the two lane halves: channels {j, 4 + j});
//   * conversion and transform in f32 (no fp16 range, no saturation, no per-cout weight scale: the table holds ones);
//   * the folded 1x1 shortcut (E2) is an f32 GEMM on the raw f32 shortcut input; the residual input arrives as 2 x 16 bytes per
//     thread and pass through ONE LDS buffer; the output sweep stores 2 x 16 bytes per lane.
// Error against the f64 convolution: ~1e-6 (the transforms' |B^T| row sums of 10 and |A^T| of up to 8 on f32 roundings).
#include <type_traits>

#include "conv_common.h"

namespace {

using namespace fdconv;

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

constexpr int cmax(int a, int b) { return a > b ? a : b; }

// ---- geometry -------------------------------------------------------------------------------------------------------------
constexpr int NTH = 512, TH = 16, TW = 16, HH = TH + 2, HW = TW + 2, BN = 256;
constexpr int CK = 8;                               // f32 channels per K chunk (one 32-byte LDS row: two 16-byte halves of 4 floats)
constexpr int SLAB = BN * 32;                       // 8 KiB: the weights of one (chunk, xi, dy) step, [256 couts][8 ch] f32
constexpr int RSLOT = 2048, NRING = 6;              // per-wave ring: 6 slots of 64 couts x 32 B (18 % NRING == 0)
static_assert(18 % NRING == 0, "the two-chunk loop body turns the ring a whole number of times");
constexpr int RING_BYTES = 8 * NRING * RSLOT;       // 96 KiB
constexpr int VXI = HH * 4 * 32;                    // one position plane: [18 halo rows][4 tiles][32 B]
constexpr int V_BYTES = 6 * VXI;                    // 13824
constexpr int V_OFF = RING_BYTES;
constexpr int Z_OFF = V_OFF + 2 * V_BYTES;
constexpr int ZROW = HW + 4;                        // z row: halo column c at position c + (c >> 2) (one pad slot per four columns), 22 slots
constexpr int ZPLANE = HH * ZROW * 16;              // 6336 = 64 (mod 128): the two channel-half planes are 16 banks apart
constexpr int Z_BYTES = 2 * ZPLANE;                 // activated halo, [channel half 2][18 rows][22 slots][16 B]: conflict-free for the
                                                    // wave-uniform-half stores of the conversion AND the word-wise reads of the transform
constexpr int RAW_OFF = Z_OFF + Z_BYTES;            // raw halo (f32) of a chunk PAIR as the DMA delivers it: 64 B per pixel = [chunk parity][half][16 B],
constexpr int RAW_PASSES = 3;                       // piece s = 4 * pixel + 2 * parity + half at s * 16; 1296 pieces, padded to 3 passes of 512 lanes
constexpr int RAW_BYTES = RAW_PASSES * NTH * 16;    // 24576
constexpr int MAIN_BYTES = RAW_OFF + RAW_BYTES;     // 163200
// epilogue: exchange buffer [wave][plane][4][64 lanes x 16 B] = the staging of one round (128 pixels x 128 couts f32, padded rows)
// in the same bytes (a barrier apart), one residual buffer (one round: 4 passes x 512 threads x 32 B), bias, statistics
constexpr int X_BYTES = 8 * 8192;                   // exchange buffer (two of them: one barrier per round)
constexpr int S_PITCH = 128 * 4 + 16;
constexpr int S_BYTES = 128 * S_PITCH;              // 67584; two staging buffers unless the residual buffers are in use
constexpr int SK_OFF = S_BYTES, SK_BYTES = 8 * NTH * 16;   // ONE residual buffer: 4 passes x 2 pieces x 512 threads x 16 B (f32: 32 B per thread and pass)
constexpr int SCK = 2, SCD = 3;                     // shortcut GEMM: K steps (8 channels) per stage, stages resident (SCD - 1 in flight)
constexpr int WS_BYTES = SCK * SLAB;                // shortcut weights of a stage: [K step][cq][ct][32 couts][32 B], shared by the workgroup
constexpr int XS_OFF = SCD * WS_BYTES, XS_BYTES = SCK * 8192;   // SCD weight stages below, SCD x stages here
constexpr int BIAS_OFF = 2 * S_BYTES;               // = 135168: [256] f32; above what E1 (2 X) and E3 (2 S, or S + SK) use.  E2 (shortcut:
                                                    // buffers up to XS_OFF + SCD * XS_BYTES) overlaps it: the table is written after E2
static_assert(BIAS_OFF >= 2 * X_BYTES && BIAS_OFF >= SK_OFF + SK_BYTES && XS_OFF + SCD * XS_BYTES <= 160 * 1024, "epilogue LDS map");
constexpr int ISC_OFF = BIAS_OFF + 1024;            // [256] f32: the packed format's per-cout scale table (ones for f32 weights) x `scale`
constexpr int STAT_OFF = ISC_OFF + 1024;            // [2][8 waves][16 octets][16] f32 = 16 KiB
constexpr int LDS_BYTES = cmax(MAIN_BYTES, STAT_OFF + 16384);
constexpr int NPIECE = HH * HW * 4;                 // 1296 raw pieces of 16 B (4 channels) per chunk pair
static_assert(LDS_BYTES <= 160 * 1024 && NPIECE <= RAW_PASSES * NTH && HH * HW <= 2 * 256, "LDS layout");

// LDS-DMA from inline asm: not counted by hipcc, every wait for it is an explicit counted s_waitcnt vmcnt(N) below.  Each lane's 16 bytes
// come from (wave-uniform base + per-lane 32-bit byte offset) and land at LDS byte (M0 + instruction offset + lane * 16).  M0 is written
// in the same statement that uses it (cdna_hip_programming.md 5.7) and NOT restored: nothing else in this file uses M0 (gfx9+ LDS
// instructions do not).  M0 cannot go on the clobber list: hipcc treats it as a reserved register ("clobbering them may lead to undefined
// behaviour", -Winline-asm), so the BUILD checks the ISA instead -- flowdec_amd/build.py: no M0 use outside these statements, no scratch.
__device__ __forceinline__ void glds16s(const void* sbase, unsigned voff, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
// two consecutive 1-KiB pieces (global and LDS both + 1024 for the second)
__device__ __forceinline__ void glds16s_x2(const void* sbase, unsigned voff, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024" ::"v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
}

template <bool ACT, bool SKIP, bool SC>
__global__ __launch_bounds__(NTH, 2) void conv_wino4f_kernel(ConvArgs p) {
  static_assert(!(SKIP && SC), "residual input and folded shortcut exclude each other in this kernel");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  FD_T2(const unsigned long long t2_entry = __builtin_amdgcn_s_memtime();)

  // ---- tile decode with XCD-aware remap (as conv_mfma.hip) ------------------------------------------------------------
  const int bid = blockIdx.x, nblk = gridDim.x;
  int lid;
  {
    const int xcd = bid & 7, qq = nblk >> 3, rr = nblk & 7;
    lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  }
  if (p.reversed) lid = nblk - 1 - lid;   // FD_TILE_REVERSED: start where the producing launch stopped (its last lines are still cached)
  int pt = lid;
  const int tw_i = pt % p.tiles_w; pt /= p.tiles_w;
  const int th_i = pt % p.tiles_h;
  const int b = pt / p.tiles_h;
  const int h0 = th_i * TH, w0 = tw_i * TW;
  const int H = p.H, W = p.W;

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int cq = wave & 3, xt = wave >> 2;

  // ---- halo loader.  The halo of a chunk PAIR (16 f32 channels) is requested at once: 324 pixels x 64 contiguous bytes, FOUR adjacent
  // lanes per pixel (one memory request per pixel and pair: the 32-byte gathers of a per-chunk request were a third of the vector L1's
  // time), three passes of 512 lanes, straight to LDS by DMA (RAW: piece 4 pixel + 2 parity + half at 16 bytes each, pass i of wave w
  // at (i * 8 + w) KiB in lane order).  CONVERSION, per chunk: wave w converts channel half hq = w & 1 of the pixels hp = i * 256 +
  // (w >> 1) * 64 + lane in pass i = 0, 1 (pass 1: 68 pixels, waves 0..3 only) -- the half is WAVE-UNIFORM, so the GroupNorm affine
  // (a, d) of the wave's 4 channels is 8 scalar registers (one s_load_dwordx8 per chunk: no LDS table); the lane reads the piece back
  // (written by another lane's request), activates it and stores it into z -- nothing is held in registers while the load is in
  // flight.  The K loop is bound by vector-instruction issue as soon as the per-chunk work costs more than a few hundred instructions
  // per wave (SQ counters, profiles/r04_wino4_*): everything a lane needs per chunk is computed ONCE here and kept in registers. -----
  auto opaque = [&](int v) { asm volatile("" : "+v"(v)); return v; };
  const size_t img_elems = (size_t)H * W;
  // the (at most two) concat segments in scalar registers: no kernel-argument loads inside the K loop
  const float* const sb0 = reinterpret_cast<const float*>(p.seg[0].src) + (size_t)b * img_elems * p.seg[0].C;
  const bool two3 = p.nseg > 1 && p.seg[1].taps == 9;   // (segments with taps == 1 are the folded shortcut: epilogue phase E2)
  const float* const sb1 = two3 ? reinterpret_cast<const float*>(p.seg[1].src) + (size_t)b * img_elems * p.seg[1].C : sb0;
  const int sC0 = p.seg[0].C, sC1 = two3 ? p.seg[1].C : 0;
  const int nch0 = sC0 / CK, n3 = nch0 + sC1 / CK;
  const int hq = wave & 1;
  const bool pass1 = wave < 4;       // (the other waves request a duplicate in pass 1 -- every wave has the same number of
                                     //  vector-memory operations in flight -- and skip its conversion)
  // per-lane halo constants.  Ordering between the request side (lane t, piece t + 512 i) and the conversion side (piece (hp * 4 + 2 e +
  // hq), written by another lane): "request after the barrier of step 3 of an even chunk (every conversion of the previous pair is done),
  // landed before the barrier of step 8 (every wave waits for its own requests first), converted after it".  Pixels outside the image:
  // the request reads pixel 0 (harmless), the conversion masks the value.  Lanes without a pixel in pass 1 redo pass 0.
  int hpix[RAW_PASSES], zadr[2];
  unsigned hvalid = 0;
#pragma unroll
  for (int i = 0; i < RAW_PASSES; ++i) {   // request side: piece sl = t + 512 i -> pixel sl >> 2 (four adjacent lanes share a pixel's 64 bytes)
    const int sl = t + i * NTH;
    const int hp = sl < NPIECE ? sl >> 2 : 0;
    const int hr = hp / HW, hc = hp - hr * HW;
    const int gh = h0 - 1 + hr, gw = w0 - 1 + hc;
    hpix[i] = (gh >= 0 && gh < H && gw >= 0 && gw < W) ? gh * W + gw : 0;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {            // conversion side
    const int base = ((t >> 7) << 6) + lane;
    const bool has = i == 0 || base + 256 < HH * HW;
    const int hp = has ? base + i * 256 : base;
    const int hr = hp / HW, hc = hp - hr * HW;
    const int gh = h0 - 1 + hr, gw = w0 - 1 + hc;
    const bool ok = gh >= 0 && gh < H && gw >= 0 && gw < W;
    zadr[i] = (hp * 4 + hq) * 16 + ((hq * ZPLANE + (hr * ZROW + hc + (hc >> 2)) * 16) << 16);   // low half: RAW piece offset (parity 0), high half: z offset
    if (ok) hvalid |= 1u << i;
  }
  // state of the chunk whose halo is in flight / being converted (wave-uniform)
  const float* nbase = sb0;
  int nC2 = 0, ncb = 0;              // bytes per pixel, byte offset of the chunk's 8 channels inside a pixel
  int cnext = 0;                     // index of the next chunk to request
  f32x8 aff;                         // (a, d) x 4 channels (this wave's half) of the chunk being converted (scalar registers)
  const float* const affp = ACT ? p.affine + ((size_t)b * p.affC + hq * 4) * 2 : nullptr;
  auto next_chunk = [&]() {          // (past the last chunk the state stays: load_halo turns the request into a harmless re-read)
    if (cnext < n3) {
      const bool first = cnext < nch0;
      const int cch = first ? cnext : cnext - nch0;
      nbase = first ? sb0 : sb1;
      nC2 = (first ? sC0 : sC1) * 4;
      ncb = cch * CK * 4;
      if constexpr (ACT)             // affine table = [C0 + C1] pairs in concat order (fd_conv2d: aff_off = 0 / C0)
        asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(aff) : "s"(affp + (size_t)cnext * CK * 2) : "memory");
    }
    ++cnext;
  };
  auto aff_wait = [&]() {            // the scalar load is invisible to hipcc: wait for it (and re-define the registers) before the first use
    if constexpr (ACT) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(aff)::"memory");
  };
  // past the end of the K loop every lane re-reads element 0 of the last tensor (one cache line; the data is never used)
  const unsigned rawdst = (unsigned)(RAW_OFF + wave * 1024);
  auto load_halo = [&]() {                // the 16 channels of the chunk pair that starts at the chunk next_chunk has just set up
    const int on = cnext <= n3 ? 1 : 0;   // (next_chunk has already counted this request)
    const int h16 = (opaque(t) & 3) * 16;
#pragma unroll
    for (int i = 0; i < RAW_PASSES; ++i) glds16s(nbase, (unsigned)((hpix[i] * nC2 + ncb + h16) * on), rawdst + i * 8192);
  };
  // one slot: RAW (f32, 4 channels) -> [silu(a x + d)] -> z, zero padding AFTER the activation (an AND: no branch)
  auto conv_slot = [&](int i, int e) {   // e = parity of the chunk inside its pair (compile-time)
    const int za = opaque(zadr[i]);
    const u32x4 raw = *reinterpret_cast<const u32x4*>(smem + RAW_OFF + e * 32 + (za & 0xffff));
    const unsigned vm = ((hvalid >> i) & 1u) ? 0xffffffffu : 0u;
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned ur = raw[j];   // (element -> scalar first: see mma below)
      float x = __builtin_bit_cast(float, ur);
      if constexpr (ACT) x = fd_silu(fmaf(x, aff[2 * j], aff[2 * j + 1]));
      o[j] = __builtin_bit_cast(unsigned, x) & vm;
    }
    *reinterpret_cast<u32x4*>(smem + Z_OFF + ((unsigned)za >> 16)) = o;
  };

  // ---- input transform z -> V planes, one channel (one 32-bit word) per lane: 6 words in, 6 words out.  Lane l of a 32-lane group
  // takes word l & 7 (channel) of tile l >> 3 of ONE halo row: the group's reads cover four 16-byte slots per channel half that
  // the z layout keeps in distinct banks, its stores 128 contiguous bytes of a V plane -- no bank conflicts (the first version's
  // item order cost 430 LDS cycles per chunk in conflicts, SQ_LDS_BANK_CONFLICT).  Halo rows 0..15 = the sixteen 32-lane groups of
  // the workgroup (rows 0..7 wave group 0, rows 8..15 wave group 1), rows 16, 17 = the two halves of wave 1 one step later.
  auto item_z = [&](int hrow, int wt, int word) { return Z_OFF + (word >> 2) * ZPLANE + (hrow * ZROW + 5 * wt) * 16 + (word & 3) * 4; };
  auto item_v = [&](int hrow, int wt, int word) {
    const int tidx = hrow * 4 + wt;
    return V_OFF + tidx * 32 + (((word >> 2) ^ ((tidx >> 3) & 1)) * 16) + (word & 3) * 4;
  };
  const int tz0 = item_z(t >> 5, (t >> 3) & 3, t & 7), tv0 = item_v(t >> 5, (t >> 3) & 3, t & 7);
  auto transform_at = [&](int za, int va, int vbuf) {
    // (opaque: hipcc otherwise hoists one address register per plane and buffer out of the K loop -- twelve registers it does not have)
    const char* const zp = smem + opaque(za);
    char* const vb = smem + opaque(va) + vbuf * V_BYTES;
    float z_[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) z_[j] = *reinterpret_cast<const float*>(zp + (j + (j >> 2)) * 16);   // halo column 4 wt + j
    const float z0 = z_[0], z1 = z_[1], z2 = z_[2], z3 = z_[3], z4 = z_[4], z5 = z_[5];
    const float t1 = fmaf(z2, -4.f, z4);   // z4 - 4 z2
    const float t2 = fmaf(z1, -4.f, z3);   // z3 - 4 z1
    const float t3 = z4 - z2, t4 = z3 - z1;
    // plane slots in the order xi = 0 1 2 | 5 3 4 (see the epilogue)
    *reinterpret_cast<float*>(vb + 0 * VXI) = fmaf(z0, 4.f, fmaf(z2, -5.f, z4));   // xi = 0
    *reinterpret_cast<float*>(vb + 1 * VXI) = t1 + t2;                             // xi = 1
    *reinterpret_cast<float*>(vb + 2 * VXI) = t1 - t2;                             // xi = 2
    *reinterpret_cast<float*>(vb + 4 * VXI) = fmaf(t4, 2.f, t3);                   // xi = 3
    *reinterpret_cast<float*>(vb + 5 * VXI) = fmaf(t4, -2.f, t3);                  // xi = 4
    *reinterpret_cast<float*>(vb + 3 * VXI) = fmaf(z1, 4.f, fmaf(z3, -5.f, z5));   // xi = 5
  };
  const bool extra_wave = wave == 1;
  auto transform_extra = [&](int vbuf) {
    if (extra_wave) transform_at(item_z(16 + (lane >> 5), (lane >> 3) & 3, lane & 7), item_v(16 + (lane >> 5), (lane >> 3) & 3, lane & 7), vbuf);
  };

  // ---- weight stream: this wave's 2 KiB (cout blocks ct = 0, 1: two consecutive 1-KiB pieces) of its step f -> ring slot f % NRING.
  // packed layout: [chunk][xt][xi][dy][cq][ct][32 couts][32 B]; the wave's 9 steps of a chunk are 9 consecutive slabs.  The stream
  // runs up to a ring length past the wave's last step: fd_wino4f_packed_bytes pads the buffer by one chunk (what lands is never read).
  const char* wp = reinterpret_cast<const char*>(p.w) + (size_t)xt * 9 * SLAB + cq * 2048;   // first step of the current chunk pair
  asm volatile("" : "+s"(wp));
  const unsigned lane16 = (unsigned)(lane * 16);
  const unsigned wring = (unsigned)(wave * NRING * RSLOT);
  auto dma_step = [&](int f, int slot) {   // f = step index relative to the current chunk pair (compile-time)
    glds16s_x2(wp + (f / 9) * 18 * SLAB + (f % 9) * SLAB, lane16, wring + slot * RSLOT);
  };

  // ---- per-lane fragment coordinates --------------------------------------------------------------------------------------
  const int l31 = lane & 31, lh = lane >> 5;
  const int wa_lane = (int)wring + l31 * 32 + ((lh ^ ((l31 >> 3) & 1)) * 16);   // A: row l31 of the slot, cout block ct at + 1024
  // B: tile l31 (+ 32 at + 1024), halo row shift dy: row idx = l31 + 4 dy at idx * 32, half lh ^ ((idx >> 3) & 1).  dy = 2 flips the half
  // of every lane (idx + 8), dy = 1 that of the lanes with l31 & 4 (the carry into bit 3)
  const int vb0 = V_OFF + xt * 3 * VXI /* plane slots 0 1 2 | 5 3 4 */ + l31 * 32 + ((lh ^ ((l31 >> 3) & 1)) * 16);
  const int vb1 = (vb0 ^ ((l31 & 4) << 2)) + 128;

  f32x16 acc[3][2][2];
#pragma unroll
  for (int x = 0; x < 3; ++x)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[x][ct][nt][e] = 0.f;

  auto lds_wait = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
  auto barrier = [&]() {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  // Fragments: three quads per operand.  Step k multiplies A0 = qa[2k % 3], A1 = qa[(2k + 1) % 3] with B0 = qb[2k % 3], B1 = qb[(2k + 1) % 3]
  // in the order (A0,B0) (A0,B1) (A1,B0) (A1,B1); the spare quads (2k + 2) % 3 receive A0 / B0 of step k + 1 at the start of step k,
  // A1 of step k + 1 goes into A0's quad after the second MFMA, B1 of step k + 1 into B0's quad after the third (so the roles turn as
  // (A0, A1, spare) -> (spare, A0, A1)): every fragment is requested two to four MFMAs before its first use with 24 registers.
  u32x4 qa[3], qb[3];
  auto mma = [&](int xl, int ct, int nt, const u32x4& a, const u32x4& bq) {   // 8 channels = four k = 2 matrix instructions (channels {j, 4 + j})
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // (vector element -> scalar FIRST: __builtin_bit_cast applied directly to an ext-vector element lvalue reads element 0 for every
      //  index with hipcc / ROCm 7.2 -- conv_mfma.hip act_slot)
      const unsigned ua = a[j], ub = bq[j];
      acc[xl][ct][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, ua), __builtin_bit_cast(float, ub), acc[xl][ct][nt], 0, 0, 0);
    }
  };
  auto rd = [&](int off) { return *reinterpret_cast<const u32x4*>(smem + off); };
  // LDS offsets of the fragments of step k18 (0..17 inside the chunk pair; 18 = step 0 of the next pair)
  auto a_off = [&](int k18, int ct) { return wa_lane + (k18 % NRING) * RSLOT + ct * 1024; };
  auto b_off = [&](int k18, int nt) {
    const int s = k18 % 9, half = (k18 / 9) & 1;
    const int dy = s % 3;
    return (dy == 0 ? vb0 : dy == 1 ? vb1 : (vb0 ^ 16) + 256) + half * V_BYTES + (s / 3) * VXI + nt * 1024;
  };

  // ---- prologue: first halo, the first ring of weight steps and the affine of chunk 0 in one memory round trip ----------------------
  next_chunk();
  load_halo();
#pragma unroll
  for (int k = 0; k < NRING; ++k) dma_step(k, k);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  aff_wait();
  barrier();                // every lane's halo slots have landed
  conv_slot(0, 0);
  if (pass1) conv_slot(1, 0);
  next_chunk();             // (chunk 1: its halo arrived with chunk 0's)
  lds_wait();
  barrier();                // z complete
  transform_at(tz0, tv0, 0);
  transform_extra(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  lds_wait();
  barrier();
  aff_wait();
  if (xt == 0) conv_slot(0, 1);   // wave group 0 is one step ahead with its conversions (see below)
  FD_T2(const unsigned long long t2_first = __builtin_amdgcn_s_memtime();)
  qa[0] = rd(a_off(0, 0)); qb[0] = rd(b_off(0, 0)); qb[1] = rd(b_off(0, 1)); qa[1] = rd(a_off(0, 1));

  // ---- K loop.  Chunk c multiplies V[c & 1]; meanwhile the halo of chunk c + 1 (in RAW) is activated and stored into z, transformed
  // into V[(c + 1) & 1], and the halo of chunk c + 2 is requested.  z may be written between the barrier of step 8 (all transforms of
  // the previous chunk have read it) and the barrier of step 3; the transforms run between the barrier of step 3 and that of step 8.
  // The two waves of a SIMD (groups xt = 0 / 1) do this vector work in DIFFERENT steps:
  //   group 0: pass 0 converted in step 8 (of the previous chunk), pass 1 in step 0, transform of halo rows 0..7 in step 4, of rows
  //            16, 17 in step 5 (wave 1)
  //   group 1: conversions in steps 1 and 2, transform of halo rows 8..15 in step 5
  //   both:    next affine requested in step 3 of every chunk, the halo of the next chunk PAIR in step 3 of the even chunks
  // A step starts with a counted wait for the weights of the NEXT step (its first fragments are requested right away); in flight on
  // this wave's vector-memory counter at that wait, oldest first: the weight pieces of steps s + 1 .. s + 5 (2 each) and, in the four
  // steps after a halo request, its three pieces: vmcnt(8 / 11).  The weights of step s + 6 are requested after the third MFMA of step
  // s: by then both weight fragments of step s have arrived and its ring slot is free.
  // Two chunks per iteration: 18 steps = three turns of the ring = six turns of the fragment quads, and the V buffers swap back, so
  // that every LDS offset is an immediate and the loop has ONE set of MFMA sites (two bodies in one loop double the accumulators).
  constexpr int NW = 2 * (NRING - 2);   // weight pieces that may stay in flight at the wait
  for (int c = 0; c < n3; c += 2) {
    int xg = xt;                       // (opaque per iteration: hipcc otherwise unswitches the loop on the wave group -- two loop bodies,
    asm volatile("" : "+s"(xg));       //  two sets of MFMA sites, spilled accumulators)
#pragma clang loop unroll(full)
    for (int k = 0; k < 18; ++k) {
      const int s = k % 9, half = k / 9;
      const int vn = half ^ 1;
      const int xl = s / 3;
      u32x4 &A0 = qa[(2 * k) % 3], &A1 = qa[(2 * k + 1) % 3], &A2 = qa[(2 * k + 2) % 3];
      u32x4 &B0 = qb[(2 * k) % 3], &B1 = qb[(2 * k + 1) % 3], &B2 = qb[(2 * k + 2) % 3];
      // the halo request of step 3 (issued BEFORE that step's weight request) is younger than the awaited weights in steps 4 .. 7 and
      // older than those of step 8: that wait also covers it, and the barrier behind it publishes the RAW slots
      if (s >= 4 && s <= 7 && half == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NW + RAW_PASSES) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NW) : "memory");
      if (s == 3 || s == 8) { lds_wait(); barrier(); }   // s == 3: z complete;  s == 8: V[vn] complete, every wave has its last V[vc] fragments
      A2 = rd(a_off(k + 1, 0));
      B2 = rd(b_off(k + 1, 0));
      __builtin_amdgcn_sched_barrier(0);
      mma(xl, 0, 0, A0, B0);
      mma(xl, 0, 1, A0, B1);
      __builtin_amdgcn_sched_barrier(0);
      A0 = rd(a_off(k + 1, 1));
      mma(xl, 1, 0, A1, B0);
      __builtin_amdgcn_sched_barrier(0);
      B0 = rd(b_off(k + 1, 1));
      if (s == 3) { next_chunk(); if (half == 0) load_halo(); }   // (after the barrier of this step: every conversion of the previous pair is done)
      dma_step(k + NRING, k % NRING);              // step s + 6 into the slot whose fragments have both arrived
      mma(xl, 1, 1, A1, B1);
      __builtin_amdgcn_sched_barrier(0);
      // the chunk's vector work, while the quads A1 / B1 are dead
      // (step 8 converts the chunk two ahead: same parity as the current one; steps 0 .. 2 the next chunk: the other parity)
      if (s == 8) { if (xg == 0) { aff_wait(); conv_slot(0, half); } }
      if (s == 0) { if (xg == 0 && pass1) conv_slot(1, half ^ 1); }
      if (s == 1) { if (xg == 1) { aff_wait(); conv_slot(0, half ^ 1); } }
      if (s == 2) { if (xg == 1 && pass1) conv_slot(1, half ^ 1); }
      if (s == 4) { if (xg == 0) transform_at(tz0, tv0, vn); }
      if (s == 5) { if (xg == 1) transform_at(tz0, tv0, vn); else transform_extra(vn); }
      __builtin_amdgcn_sched_barrier(0);
    }
    wp += 36 * SLAB;
    asm volatile("" : "+s"(wp));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // all fragment reads / DMA done before the epilogue reuses the LDS
  FD_T2(const unsigned long long t2_loop = __builtin_amdgcn_s_memtime();)

  // ---- epilogue -------------------------------------------------------------------------------------------------------
  // Lane (l31, lh) of wave (cq, xt) holds, for cout block ct and tile block nt, acc[x][ct][nt][e]: cout ct * 128 + cq * 32 + 8 (e >> 2)
  // + 4 lh + (e & 3) of tile nt * 32 + l31 (row nt * 8 + (l31 >> 2), pixels 4 (l31 & 3) .. + 3); x = 0, 1, 2 are the positions
  // 0, 1, 2 (xt = 0) or 5, 3, 4 (xt = 1).
  //   E1  output transform.  Per (ct, nt): both wave groups run the same code with wave-uniform coefficients (a branch costs spilled
  //       accumulator tiles):  s = m1 + m2, d = m1 - m2;  send0 = s, send1 = ke d;  ya = ka m0 + ks s + recv0,  yb = kb m0 + kd d + recv1
  //         xt = 0 (m = M0, M1, M2): sends a2 = s, a1 = d;   y0 = M0 + s + b0,  y1 = d + b1             (ka ks kb kd ke = 1 1 0 1 1)
  //         xt = 1 (m = M5, M3, M4): sends b0 = s, b1 = 2 d;  y2 = 4 s + a2,     y3 = M5 + 8 d + a1      (ka ks kb kd ke = 0 4 1 8 2)
  //       the two waves of a cout block swap two planes through LDS (double-buffered: one barrier per round); afterwards wave (cq, xt)
  //       holds the pixel planes j = 2 xt (in m0) and 2 xt + 1 (in m1) of its couts and all 64 tiles.
  //   E2  (SC) the folded 1x1 shortcut conv Conv_2 (layerspp.py:278-279) as an f32 GEMM straight into those planes: y[cout][pixel] +=
  //       W[cout][k] x[k][pixel], v_mfma_f32_32x32x2_f32 on the raw residual stream.
  //   E3  per (ct, nt): stage the round's 128 pixels x 128 couts as [pixel plane j][tile][cout] f32, sweep with 8 couts (32 B)
  //       per lane: bias / residual / scale / statistics / store.
  float* const out = reinterpret_cast<float*>(p.out) + (size_t)b * img_elems * p.Cout;
  const float* const skip = SKIP ? reinterpret_cast<const float*>(p.skip) + (size_t)b * img_elems * p.Cout : nullptr;
  float* const biast = reinterpret_cast<float*>(smem + BIAS_OFF);      // [256] f32 (zeros without a bias)
  float* const statt = reinterpret_cast<float*>(smem + STAT_OFF);      // [ct][wave][oct 16][16] f32 partial sums
  float* const isct = reinterpret_cast<float*>(smem + ISC_OFF);        // [256] f32: 1 / (the cout's weight scale)
  auto load_bias_table = [&]() {
    if (t < BN) {
      // out = scale (acc / wscale + skip + bias) = acc (scale / wscale) + skip scale + bias scale: both tables carry `scale`, so that the
      // sweep is one fma per output pair (two with a residual) instead of fma + multiply (round 5: -0.4 % per launch)
      biast[t] = p.bias ? p.bias[(size_t)(p.bias_rows > 1 ? b : 0) * p.Cout + t] * p.scale : 0.f;
      isct[t] = p.w_scale[t] * p.scale;
    }
  };
  if constexpr (!SC) load_bias_table();   // (published by the barriers of the first round; with a shortcut: after E2, whose buffers overlap it)
  // element offset of pass ps of round (ct, nt) for this thread: staged pixel pp + 32 ps = plane ps of tile pp, cout octet oct
  auto out_off = [&](int tt, int ct, int nt, int ps) {
    const int oct = tt & 15, pp = tt >> 4;
    const int gh = h0 + nt * 8 + (pp >> 2), gw = w0 + 4 * (pp & 3) + ps;
    return (gh * W + gw) * p.Cout + ct * 128 + oct * 8;
  };
  // residual of round r -> SK by DMA: thread t requests exactly the 2 x 16 bytes (8 floats) it adds in the sweep (piece h of pass ps at
  // ((ps * 2 + h) * 512 + t) * 16), so the only synchronisation is this wave's own vmcnt.  ONE buffer: round r + 1 is requested right
  // after the thread's own reads of round r (the f32 kernel is MFMA-bound by an order of magnitude; the round trip is not hidden)
  auto skip_dma = [&](int r) {
    if constexpr (SKIP) {
      const int tt = opaque(t);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this thread's reads of the previous round have returned
#pragma unroll
      for (int ps = 0; ps < 4; ++ps)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          glds16s(skip, (unsigned)(out_off(tt, r >> 1, r & 1, ps) * 4 + h * 16), (unsigned)(SK_OFF + (ps * 2 + h) * NTH * 16 + wave * 1024));
    }
  };
  skip_dma(0);
  FD_T2(
  unsigned long long t2_r[5] = {0, 0, 0, 0, 0};
  t2_r[0] = __builtin_amdgcn_s_memtime();
  )

  const float ka = xt ? 0.f : 1.f, ks = xt ? 4.f : 1.f, kb = xt ? 1.f : 0.f, kd = xt ? 8.f : 1.f, kq = xt ? 0.25f : 1.f;
  // E1 of one round: partial transform, send, barrier, receive + combine (ya -> m0, yb -> m1); xoff = byte offset of the exchange buffer
  auto e1_round = [&](int rd_, int xoff) {
    const int ct = rd_ >> 1, nt = rd_ & 1;
    f32x16 &m0 = acc[0][ct][nt], &m1 = acc[1][ct][nt], &m2 = acc[2][ct][nt];
    const f32x2 kd2 = {kd, kd}, kq2 = {kq, kq}, ka2 = {ka, ka}, ks2 = {ks, ks}, kb2 = {kb, kb};
    char* const xmine = smem + xoff + wave * 8192 + lane * 16;
    const char* const xpeer = smem + xoff + (wave ^ 4) * 8192 + lane * 16;
    // (two channels per instruction: v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 on consecutive accumulator registers)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x2 snd[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int e = 4 * g + 2 * h;
        const f32x2 a1 = {m1[e], m1[e + 1]}, a2 = {m2[e], m2[e + 1]};
        const f32x2 s_ = a1 + a2, d_ = (a1 - a2) * kd2;
        m2[e] = s_[0]; m2[e + 1] = s_[1];
        m1[e] = d_[0]; m1[e + 1] = d_[1];
        snd[h] = d_ * kq2;                                     // ke d = (ke / kd) kd d
      }
      *reinterpret_cast<f32x4*>(xmine + g * 1024) = f32x4{m2[4 * g], m2[4 * g + 1], m2[4 * g + 2], m2[4 * g + 3]};
      *reinterpret_cast<f32x4*>(xmine + 4096 + g * 1024) = f32x4{snd[0][0], snd[0][1], snd[1][0], snd[1][1]};
    }
    lds_wait();
    barrier();
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 r0 = *reinterpret_cast<const f32x4*>(xpeer + g * 1024), r1 = *reinterpret_cast<const f32x4*>(xpeer + 4096 + g * 1024);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int e = 4 * g + 2 * h;
        const f32x2 mm = {m0[e], m0[e + 1]}, ss = {m2[e], m2[e + 1]}, dd = {m1[e], m1[e + 1]};
        const f32x2 q0 = {r0[2 * h], r0[2 * h + 1]}, q1 = {r1[2 * h], r1[2 * h + 1]};
        const f32x2 ya = __builtin_elementwise_fma(ka2, mm, __builtin_elementwise_fma(ks2, ss, q0));
        const f32x2 yb = __builtin_elementwise_fma(kb2, mm, dd + q1);
        m0[e] = ya[0]; m0[e + 1] = ya[1];
        m1[e] = yb[0]; m1[e + 1] = yb[1];
      }
    }
  };
  // E3 of one round: stage into sbuf (byte offset soff), barrier, sweep
  float ssum[8], ssq[8];
  auto e3_round = [&](int rd_, int soff) {
    const int ct = rd_ >> 1, nt = rd_ & 1;
    f32x16 &m0 = acc[0][ct][nt], &m1 = acc[1][ct][nt];
    char* const sbuf = smem + soff;
    if (nt == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) ssum[j] = ssq[j] = 0.f;
    }
    {
      char* const sbase = sbuf + ((2 * xt) * 32 + l31) * S_PITCH + (cq * 32 + 4 * lh) * 4;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        *reinterpret_cast<f32x4*>(sbase + g * 32) = f32x4{m0[4 * g], m0[4 * g + 1], m0[4 * g + 2], m0[4 * g + 3]};
        *reinterpret_cast<f32x4*>(sbase + 32 * S_PITCH + g * 32) = f32x4{m1[4 * g], m1[4 * g + 1], m1[4 * g + 2], m1[4 * g + 3]};
      }
    }
    lds_wait();
    barrier();
    __builtin_amdgcn_sched_barrier(0);
    // sweep in two batches of two passes: computed into registers first, stores back to back afterwards.  The residual of this round
    // is in SK (requested after the previous round's sweep).
    if constexpr (SKIP) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's residual pieces have landed
    {
      const int tt = opaque(t);
      const int oct = tt & 15, pp = tt >> 4;
      const f32x4 bA = *reinterpret_cast<const f32x4*>(biast + ct * 128 + oct * 8), bB = *reinterpret_cast<const f32x4*>(biast + ct * 128 + oct * 8 + 4);
      const float bv[8] = {bA[0], bA[1], bA[2], bA[3], bB[0], bB[1], bB[2], bB[3]};
      // the table entry is `scale` (x the packed format's per-cout factor: one), applied inside the fma that adds the (pre-scaled) bias / residual
      const f32x4 iA = *reinterpret_cast<const f32x4*>(isct + ct * 128 + oct * 8), iB = *reinterpret_cast<const f32x4*>(isct + ct * 128 + oct * 8 + 4);
      const float iv[8] = {iA[0], iA[1], iA[2], iA[3], iB[0], iB[1], iB[2], iB[3]};
#pragma unroll
      for (int bt = 0; bt < 2; ++bt) {
        f32x4 packed[2][2];
        int ooff[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int ps = 2 * bt + q;
          ooff[q] = out_off(tt, ct, nt, ps);
          const float* sp = reinterpret_cast<const float*>(sbuf + (pp + ps * 32) * S_PITCH) + oct * 8;
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(sp), v1 = *reinterpret_cast<const f32x4*>(sp + 4);
          float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          if constexpr (SKIP) {
            const f32x4 sk0 = *reinterpret_cast<const f32x4*>(smem + SK_OFF + ((ps * 2 + 0) * NTH + tt) * 16);
            const f32x4 sk1 = *reinterpret_cast<const f32x4*>(smem + SK_OFF + ((ps * 2 + 1) * NTH + tt) * 16);
            const float sk[8] = {sk0[0], sk0[1], sk0[2], sk0[3], sk1[0], sk1[1], sk1[2], sk1[3]};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const f32x2 s2_ = {sk[2 * k], sk[2 * k + 1]}, sc2 = {p.scale, p.scale}, b2 = {bv[2 * k], bv[2 * k + 1]};
              const f32x2 i2 = {iv[2 * k], iv[2 * k + 1]}, x_ = {v[2 * k], v[2 * k + 1]};
              const f32x2 r_ = __builtin_elementwise_fma(x_, i2, __builtin_elementwise_fma(s2_, sc2, b2));
              v[2 * k] = r_[0]; v[2 * k + 1] = r_[1];
            }
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
#pragma clang fp contract(off)
            f32x2 x = {v[2 * k], v[2 * k + 1]};
            f32x2 s1 = {ssum[2 * k], ssum[2 * k + 1]}, s2 = {ssq[2 * k], ssq[2 * k + 1]};
            if constexpr (!SKIP) {
              const f32x2 b2 = {bv[2 * k], bv[2 * k + 1]}, i2 = {iv[2 * k], iv[2 * k + 1]};
              x = __builtin_elementwise_fma(x, i2, b2);
            }
            s1 += x;
            s2 = __builtin_elementwise_fma(x, x, s2);
            v[2 * k] = x[0]; v[2 * k + 1] = x[1];
            ssum[2 * k] = s1[0]; ssum[2 * k + 1] = s1[1];
            ssq[2 * k] = s2[0]; ssq[2 * k + 1] = s2[1];
          }
          packed[q][0] = f32x4{v[0], v[1], v[2], v[3]};
          packed[q][1] = f32x4{v[4], v[5], v[6], v[7]};
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          *reinterpret_cast<f32x4*>(out + ooff[q]) = packed[q][0];
          *reinterpret_cast<f32x4*>(out + ooff[q] + 4) = packed[q][1];
        }
      }
    }
    if (nt == 1 && p.stats) {
      // channel sums of this cout half over the tile: fold the four 16-lane rows of the wave (they share the octets), then one
      // record per (wave, octet): 16 floats {sum, sumsq} x 8 channels
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        ssum[j] += __shfl_xor(ssum[j], 16, 64); ssq[j] += __shfl_xor(ssq[j], 16, 64);
        ssum[j] += __shfl_xor(ssum[j], 32, 64); ssq[j] += __shfl_xor(ssq[j], 32, 64);
      }
      if (lane < 16) {
        float* rec = statt + ((ct * 8 + wave) * 16 + lane) * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(rec + 4 * j) = f32x4{ssum[2 * j], ssq[2 * j], ssum[2 * j + 1], ssq[2 * j + 1]};
      }
    }
  };

  if constexpr (SC) {
    // ---- E1 for all rounds (exchange buffers alternate: one barrier per round; the buffer of round r is written again in round r + 2,
    // and every wave has read it before it passes the barrier of round r + 1)
#pragma unroll
    for (int rd_ = 0; rd_ < 4; ++rd_) {
      e1_round(rd_, (rd_ & 1) * X_BYTES);
      __builtin_amdgcn_sched_barrier(0);
    }
    lds_wait();
    barrier();   // all exchange reads done: the LDS below is reused
    FD_T2(t2_r[1] = __builtin_amdgcn_s_memtime();)
    // ---- E2: folded shortcut ----
    {
      // K = the S0 + S1 channels of the (at most two) shortcut segments, in stages of SCK K steps (8 channels each).  A stage of x
      // is [pixel plane j][tile][64 B] (row r = j * 64 + tile = one pixel's 16 channels), filled by DMA -- the pixel order is made by
      // the SOURCE addresses.  Weights [K step][cq][ct][32 couts][32 B] f32: the two waves of a cout block (xt = 0, 1) multiply the SAME
      // rows, so a stage's 16 KiB are loaded once -- wave (cq, xt) brings K step xt of its cout block (2 KiB), everybody reads both K
      // steps after the stage's barrier (private buffers fetched every piece twice).  SCD stages are resident: while stage s is multiplied, stages s + 1 and s + 2 are in flight (one
      // stage in flight left the fp16 kernel's phase at a quarter of the matrix rate; here a stage is 64 f32 MFMAs per wave).  One
      // barrier per stage, behind a counted wait: only the requests of the NEXT stage (4 instructions) may still be in flight.
      const int nsc = p.nseg - (sC1 ? 2 : 1);   // shortcut segments follow the 3x3 segments
      const Seg q0 = p.seg[p.nseg - nsc], q1 = p.seg[p.nseg - 1];
      const int S0 = q0.C, Stot = S0 + (nsc > 1 ? q1.C : 0);
      const int nstage = Stot / (CK * SCK);
      const float* const xb0 = reinterpret_cast<const float*>(q0.src) + (size_t)b * img_elems * q0.C;
      const float* const xb1 = reinterpret_cast<const float*>(q1.src) + (size_t)b * img_elems * q1.C;
      const char* const wsc = reinterpret_cast<const char*>(p.w) + (size_t)(n3 + 1) * 18 * SLAB + cq * 2048;
      // per-lane source of the stage DMA.  A stage row is one pixel's 64 bytes (SCK = 2 K steps x 2 channel halves): FOUR adjacent
      // lanes fetch one pixel's contiguous 64 B (one memory request), lane l of DMA instruction i of wave w fills the 16 bytes at
      // ((i * 8 + w) * 64 + l) * 16: row r = that >> 2, 16-byte position q = l & 3, which holds piece (kk * 2 + half) = q ^ ((r >> 2) & 3)
      // (the XOR keeps the 64-byte-stride fragment reads free of bank conflicts)
      static_assert(SCK == 2, "a stage row = 2 K steps x 2 halves x 16 B; K step kk of the weights is loaded by wave group xt = kk");
      int xsrc[2];   // per DMA instruction: (pixel index) << 8 | byte offset of the piece inside the pixel's 64 B
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = ((i * 8 + wave) * 64 + lane) >> 2;
        const int xtile = r & 63, xj = r >> 6;
        const int piece = (lane & 3) ^ ((r >> 2) & 3);
        xsrc[i] = (((h0 + (xtile >> 2)) * W + w0 + 4 * (xtile & 3) + xj) << 8) | (piece * 16);
      }
      const unsigned lane16e = (unsigned)(lane * 16);
      // x of stage st -> x buffer `slot`, weights of stage st -> this wave's buffer `slot`; past the last stage: a harmless re-read of
      // stage 0 (every stage issues the same number of requests: the waits below count them)
      auto stage_dma = [&](int st, int slot) {
        const int ste = st < nstage ? st : 0;
        const int c0 = ste * CK * SCK;
        const bool first = c0 < S0;
        const float* const xb = first ? xb0 : xb1;
        const int Cs = first ? S0 : Stot - S0, cc = first ? c0 : c0 - S0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
          glds16s(xb, (unsigned)(((xsrc[i] >> 8) * Cs + cc) * 4 + (xsrc[i] & 255)), (unsigned)(XS_OFF + slot * XS_BYTES + i * 8192 + wave * 1024));
        glds16s_x2(wsc + (size_t)(ste * SCK + xt) * SLAB, lane16e, (unsigned)(slot * WS_BYTES + xt * SLAB + cq * 2048));
      };
      const int wa_sc = cq * 2048 + l31 * 32 + ((lh ^ ((l31 >> 3) & 1)) * 16);
      // B fragment of (K step kk, plane pl, tile block nt): row r = (2 xt + pl) * 64 + nt * 32 + l31 at r * 64, piece kk * 2 + lh at
      // position (kk * 2 + lh) ^ ((r >> 2) & 3) = ((kk * 2 + lh) ^ ((l31 >> 2) & 3))   (the row offsets are multiples of 16 rows)
      const int xb_row = XS_OFF + ((2 * xt) * 64 + l31) * 64;
      const int xb_sw = (l31 >> 2) & 3;
#pragma unroll
      for (int d = 0; d < SCD - 1; ++d) stage_dma(d, d);
      int slot = 0;
      for (int st = 0; st < nstage; ++st) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (SCD - 2)) : "memory");   // stage st has landed (this wave's part)
        lds_wait();
        barrier();                                                                  // ... everybody's; buffer (slot + SCD - 1) % SCD is free
        const int nslot = slot == 0 ? SCD - 1 : slot - 1;                           // = (slot + SCD - 1) % SCD
        stage_dma(st + SCD - 1, nslot);
        const int wo = slot * WS_BYTES, xo = slot * XS_BYTES;
        // all twelve fragments of the stage first (the third accumulator plane is dead here: 48 registers are free), the MFMAs behind
        // them with the waits the compiler counts: one exposed LDS round trip per stage instead of four (hipcc otherwise issues every
        // group of reads right in front of its MFMAs)
        u32x4 af[SCK][2], bq[SCK][2][2];   // [K step][ct] and [K step][plane][nt]
#pragma unroll
        for (int kk = 0; kk < SCK; ++kk) {
          af[kk][0] = rd(wa_sc + wo + kk * SLAB);
          af[kk][1] = rd(wa_sc + wo + kk * SLAB + 1024);
#pragma unroll
          for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) bq[kk][pl][nt] = rd(xb_row + xo + (pl * 64 + nt * 32) * 64 + (((kk * 2 + lh) ^ xb_sw) * 16));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < SCK; ++kk)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const unsigned u0 = af[kk][0][j], u1 = af[kk][1][j], ub = bq[kk][pl][nt][j];   // (element -> scalar first)
                acc[pl][0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, u0), __builtin_bit_cast(float, ub), acc[pl][0][nt], 0, 0, 0);
                acc[pl][1][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, u1), __builtin_bit_cast(float, ub), acc[pl][1][nt], 0, 0, 0);
              }
        __builtin_amdgcn_sched_barrier(0);
        slot = slot + 1 == SCD ? 0 : slot + 1;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the re-reads past the last stage
      lds_wait();
      barrier();   // all fragment reads done: the staging buffers overlap the weight buffers and the x stages
      load_bias_table();
    }
    FD_T2(t2_r[2] = __builtin_amdgcn_s_memtime();)
    // ---- E3 for all rounds (two staging buffers: one barrier per round)
#pragma unroll
    for (int rd_ = 0; rd_ < 4; ++rd_) {
      e3_round(rd_, (rd_ & 1) * S_BYTES);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    // ---- rounds of E1 + E3.  Without a residual input the exchange buffer [0, 64 KiB) and the staging buffer behind it are separate:
    // two barriers per round (the next round's exchange stores follow this round's staging barrier, its staging stores the next
    // exchange barrier, which every wave reaches only after its sweep).  With one, the two residual buffers take that room: exchange
    // and staging share the bytes, four barriers per round.
#pragma unroll
    for (int rd_ = 0; rd_ < 4; ++rd_) {
      e1_round(rd_, 0);
      if constexpr (SKIP) { lds_wait(); barrier(); }
      e3_round(rd_, SKIP ? 0 : X_BYTES);
      if constexpr (SKIP) {
        lds_wait();
        barrier();
        if (rd_ < 3) skip_dma(rd_ + 1);   // the one residual buffer is free again (every thread re-requests its own slots)
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  FD_T2(t2_r[3] = __builtin_amdgcn_s_memtime();)
  if (p.stats) {
    lds_wait();
    barrier();
    // o = 2 * channel + which = t;  channel = ct * 128 + oct * 8 + j
    const int ch = t >> 1, which = t & 1;
    const int ct = ch >> 7, oc = (ch >> 3) & 15, j = ch & 7;
    const float* src = statt + (ct * 8 * 16 + oc) * 16 + 2 * j + which;
    float a = 0.f;
#pragma unroll
    for (int w_ = 0; w_ < 8; ++w_) a += src[w_ * 256];
    const int tile = th_i * p.tiles_w + tw_i;
    p.stats[((size_t)b * p.tiles_h * p.tiles_w + tile) * p.CoutPad * 2 + t] = a;
  }
  FD_T2(
  if (p.dbg && t == 0 && bid < 8192) {
    const unsigned long long t2_end = __builtin_amdgcn_s_memtime();
    unsigned long long* d = p.dbg + (size_t)bid * 8;
    d[0] = t2_first - t2_entry; d[1] = t2_loop - t2_first; d[2] = t2_end - t2_loop;
    d[3] = t2_r[1] - t2_r[0]; d[4] = t2_r[2] - t2_r[1]; d[5] = t2_r[3] - t2_r[2]; d[6] = t2_end - t2_r[3]; d[7] = t2_r[0] - t2_loop;
  }
  )
}

// ---- weight packing: [Cout][Cin][3][3] f32 -> [chunk of 8 channels][xt][xi][dy][256 couts][32 B] f32 --------------------------------
// U_xi = G(xi) . w[dy][0..2]; a row's two 16-byte halves (channels 0-3 / 4-7 of the chunk) are XOR-swizzled by (cout >> 3) & 1.
__device__ __forceinline__ float wino4_u(const float* g, int xi) {
  const float g0 = g[0], g1 = g[1], g2 = g[2];
  switch (xi) {
    case 0: return 0.25f * g0;
    case 1: return -(g0 + g1 + g2) / 6.f;
    case 2: return (-g0 + g1 - g2) / 6.f;
    case 3: return g0 / 24.f + g1 / 12.f + g2 / 6.f;
    case 4: return g0 / 24.f - g1 / 12.f + g2 / 6.f;
    default: return g2;
  }
}

// the scale table of the fp16 kernel's packed format (header of the buffer): f32 weights need no scaling -- ones
__global__ void wino4f_ones_kernel(float* __restrict__ tab) { tab[threadIdx.x] = 1.f; }

__global__ void wino4f_pack_kernel(const float* __restrict__ w, float* __restrict__ dst, int Cout, int C0, int C1) {
  const int Cin = C0 + C1, nchunks = Cin / CK;
  const long long total = (long long)nchunks * 18 * BN * CK;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i % CK);
    long long r = i / CK;
    const int n = (int)(r % BN); r /= BN;
    const int dy = (int)(r % 3); r /= 3;
    const int slot = (int)(r % 6);                          // wave group xt = slot / 3 owns the slabs slot % 3 = 0..2
    const int xi = slot < 3 ? slot : (slot == 3 ? 5 : slot - 1);   // positions in the order 0 1 2 | 5 3 4 (see the epilogue)
    const int chunk = (int)(r / 6);
    const int c = chunk * CK + k;   // (concat segments are multiples of 16 channels: a chunk never straddles them)
    const int co = ((n >> 5) & 1) * 128 + (n >> 6) * 32 + (n & 31);   // slab row n = [cq][ct][32] holds cout ct * 128 + cq * 32 + m
    float v = 0.f;
    if (co < Cout) v = wino4_u(w + (((size_t)co * Cin + c) * 3 + dy) * 3, xi);
    const int half = (k >> 2) ^ ((n >> 3) & 1);
    dst[(i - k) + half * 4 + (k & 3)] = v;
  }
}

// shortcut weights [Cout][S] f32 -> [K step of 8 channels][slab row n = [cq][ct][32]][32 B] f32, halves swizzled by (n >> 3) & 1
__global__ void wino4f_pack_sc_kernel(const float* __restrict__ w, float* __restrict__ dst, int Cout, int S) {
  const long long total = (long long)(S / CK) * BN * CK;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i % CK);
    long long r = i / CK;
    const int n = (int)(r % BN);
    const int ks = (int)(r / BN);
    const int co = ((n >> 5) & 1) * 128 + (n >> 6) * 32 + (n & 31);
    const int half = (k >> 2) ^ ((n >> 3) & 1);
    dst[(i - k) + half * 4 + (k & 3)] = co < Cout ? w[(size_t)co * S + ks * CK + k] : 0.f;
  }
}

}  // namespace

bool fd_wino4f_supported(int Cout, int C0, int C1, int S0, int S1, int ksize) {
  return ksize == 3 && Cout == BN && C0 > 0 && C0 % 16 == 0 && C1 % 16 == 0 && S0 % 16 == 0 && S1 % 16 == 0 && (S1 == 0 || S0 > 0) &&
         (C0 + C1) * 8 <= AFF_BYTES;
}
bool fd_wino4f_shape_ok(int H, int W) { return H % TH == 0 && W % TW == 0; }

namespace {
// The packed buffer has the fp16 kernel's shape: a fixed 4-KiB header (scale table [2][256] f32: ones here), then the packed steps.
constexpr long long W4_HDR = 4096;
}  // namespace

long long fd_wino4f_packed_bytes(int Cout, int C0, int C1, int S0, int S1) {
  (void)Cout;
  // header, 3x3 part + one chunk (the weight stream runs a ring length past the last step), then the shortcut K steps (+ slack)
  return W4_HDR + (long long)((C0 + C1) / CK + 1) * 18 * SLAB + (long long)((S0 + S1) / CK) * SLAB + 16 * 1024;
}

int fd_wino4f_pack_weights(const float* w, const float* w_sc, void* packed, int Cout, int C0, int C1, int S0, int S1, hipStream_t st) {
  const long long total = (long long)((C0 + C1) / CK) * 18 * BN * CK;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  float* const tab = reinterpret_cast<float*>(packed);
  char* const steps = reinterpret_cast<char*>(packed) + W4_HDR;
  hipLaunchKernelGGL(wino4f_ones_kernel, dim3(1), dim3(2 * BN), 0, st, tab);
  hipLaunchKernelGGL(wino4f_pack_kernel, dim3(blocks), dim3(256), 0, st, w, (float*)steps, Cout, C0, C1);
  if (w_sc) {
    const long long tsc = (long long)((S0 + S1) / CK) * BN * CK;
    const int bsc = (int)((tsc + 255) / 256 > 4096 ? 4096 : (tsc + 255) / 256);
    hipLaunchKernelGGL(wino4f_pack_sc_kernel, dim3(bsc), dim3(256), 0, st, w_sc,
                       reinterpret_cast<float*>(steps + (size_t)((C0 + C1) / CK + 1) * 18 * SLAB), Cout, S0 + S1);
  }
  FD_LAUNCH_CHECK();
  return FD_OK;
}

namespace {
template <bool ACT>
int set_attr4() {
  FD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino4f_kernel<ACT, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  FD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino4f_kernel<ACT, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  FD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino4f_kernel<ACT, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  return FD_OK;
}
template <bool ACT>
void launch4(const ConvArgs& a, bool sc, dim3 grid, dim3 block, hipStream_t st) {
  if (sc) hipLaunchKernelGGL((conv_wino4f_kernel<ACT, false, true>), grid, block, LDS_BYTES, st, a);
  else if (a.skip) hipLaunchKernelGGL((conv_wino4f_kernel<ACT, true, false>), grid, block, LDS_BYTES, st, a);
  else hipLaunchKernelGGL((conv_wino4f_kernel<ACT, false, false>), grid, block, LDS_BYTES, st, a);
}
}  // namespace

int fd_wino4f_init_attributes() {
  FD_TRY(set_attr4<false>());
  FD_TRY(set_attr4<true>());
  return FD_OK;
}

int fd_wino4f_launch(ConvArgs a, hipStream_t st) {
  FD_REQUIRE(fd_wino4f_shape_ok(a.H, a.W), "fd_conv2d: FD_WINOGRAD4 (f32) needs H %% 16 == 0 and W %% 16 == 0 (got %d x %d)", a.H, a.W);
  bool sc = false;
  for (int s = 0; s < a.nseg; ++s) sc = sc || a.seg[s].taps == 1;
  a.w_scale = reinterpret_cast<const float*>(a.w);                   // header of the packed buffer (launch-independent position)
  a.w = reinterpret_cast<const char*>(a.w) + W4_HDR;                 // the packed steps
  FD_REQUIRE(!(sc && a.skip), "fd_conv2d: FD_WINOGRAD4 takes a folded shortcut or a residual input, not both");
  a.tiles_h = a.H / TH;
  a.tiles_w = a.W / TW;
  a.tiles_n = 1;
  a.CoutPad = BN;
  const long long nblk = (long long)a.B * a.tiles_h * a.tiles_w;
  FD_REQUIRE(nblk > 0 && nblk < (1ll << 31), "conv grid out of range");
  const dim3 grid((unsigned)nblk), block(NTH);
  if (a.affine) launch4<true>(a, sc, grid, block, st);
  else launch4<false>(a, sc, grid, block, st);
  FD_LAUNCH_CHECK();
  return FD_OK;
}
