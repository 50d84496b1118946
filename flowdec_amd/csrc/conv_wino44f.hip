// conv_wino44f.hip -- 3x3 convolution as 2-D Winograd F(4x4, 3x3) in EXACT FLOAT32 on the gfx950 matrix cores (round 6).
//
// Same contract as the other convolution kernels (ddpm_conv3x3, flowdec/backbones/ncsnpp_utils/layers.py:128-134, with the
// ResnetBlockBigGANpp surroundings of layerspp.py:252-284 fused in: GroupNorm+SiLU operand transform, time / conv bias, folded 1x1
// shortcut conv, residual, 1/sqrt(2), statistics of the output for the next GroupNorm, virtual channel concat) for the fp32 mode:
//   V = B^T d B (6 x 6 input tile d, stride 4),  U = G g G^T,  M_(i,j) = sum_c U_(i,j) V_(i,j) (36 products per 16 outputs),  Y = A^T M A
// i.e. 2.25 multiply-adds per output and input channel instead of 9 (direct) or 4.5 (conv_wino4f.hip: F(4,3) along W only) -- a quarter /
// half of the MFMAs.  In float32 this pays: v_mfma_f32_16x16x4_f32 runs at the f32 VECTOR rate (1/16 of the fp16 matrix rate), the launch
// is bound by the matrix pipe at any grid size.  (In fp16 the 2-D form is neither accurate -- the transforms amplify an operand rounding
// ~100 x -- nor affordable: 2.25 accumulator planes per output.)
//
//   * one workgroup = 16 x 16 output pixels (16 tiles of 4 x 4) x 128 output channels, 8 waves; wave w owns couts 16 w .. 16 w + 15 for ALL
//     36 positions and all 16 tiles: 36 accumulator tiles of 16 x 16 (v_mfma_f32_16x16x4_f32) = 144 registers, the whole output
//     transform stays inside a lane (no exchange between waves);
//   * K walks 8-channel chunks.  PRODUCER, wave-private: wave w owns the tiles (row w >> 1, columns 2 (w & 1), + 1), requests their
//     6 x 10-pixel window by LDS-DMA two chunks ahead into one of its two z windows ([channel half][pixel][4 channels]: the half is uniform
//     per request, so the GroupNorm affine of a chunk is 16 scalar registers, loaded one step ahead), converts it IN PLACE ([silu(a x + d)],
//     zero padding; a raw input inside the image needs no conversion at all) and transforms it: lanes = (channel 8, tile 2, third 3), all 36
//     inputs of a lane's 6 x 6 tile in one round of LDS reads -> V (LDS, three buffers, [position group of 4][tile][channel slot][4
//     positions]: one ds_read_b128 feeds 4 MFMAs);
//   * weights: transformed at pack time, laid out so that a lane's operand of 4 consecutive positions is one aligned 16-byte piece of
//     a contiguous 1-KiB wave piece; the pieces go from L2 into a 9-slot per-wave LDS ring by DMA, 9 steps ahead (18 pieces per chunk:
//     a step's slot is a compile-time constant); every vector-memory operation of the loop is a DMA, the s_waitcnt vmcnt(N) are exact;
//   * the folded 1x1 shortcut (Conv_2, layerspp.py:276-284) is more K chunks on the raw shortcut input: a centre-tap kernel transforms
//     to the 16 inner positions only, so a shortcut chunk costs 16 of the 36 MFMAs -- exactly the 1x1 convolution's multiply-adds;
//   * epilogue: Y = A^T M A on 4-cout vectors in registers, bias / residual (rows requested one ahead) / scale, statistics by cross-lane
//     adds, 16-byte stores.
// Error against the f64 convolution: ~1e-6 (tests/test_hip_configs.py test_conv2d_winograd4_f32[44-*]).  Per launch 1.1-1.4 x the F(4,3)
// float32 kernel and 1.8-2.2 x the direct one (profiles/r06_wino44f.txt); the K loop runs at 60-67 % of its MFMA floor (2 x 72 MFMAs x 32
// cycles per chunk and SIMD) -- the rest is the producer's vector instructions, which cannot overlap float32 MFMAs on this chip.
#include <type_traits>

#include "conv_common.h"

namespace {

using namespace fdconv;

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int NTH = 512, TH = 16, TW = 16, BN = 128, CK = 8;
constexpr int WR = 6, WC = 10, WPIX = WR * WC;       // a wave's halo window: 2 horizontally adjacent tiles of 4 x 4 outputs = 6 x 10 pixels
constexpr int ZHALF = (WPIX + 1) * 4;                // floats between the two channel halves of a window: [half][pixel 60 (+1 pad)][4 channels]
                                                     // (244 = 20 mod 32: the transform's reads of a (channel, tile) lane group hit 32 different banks)
constexpr int ZWAVE = 128 * 4;                       // floats of z per window (2 KiB; 121 slots used)
constexpr int V_FLOATS = 9 * 16 * CK * 4;            // [group of 4 positions][tile][channel slot][4] = 4608 floats (18 KiB)
constexpr int WSTEP = 18;                            // weight pieces (1 KiB) per chunk and wave: 9 position groups x 2 channel quads
constexpr int RD = 9;                                // per-wave weight ring: 9 slots of 1 KiB; 18 % RD == 0: a step's slot is a compile-time constant
static_assert(WSTEP % RD == 0 && WPIX <= 64, "ring / window layout");
constexpr int Z_OFF = 0, V_OFF = 8 * 2 * ZWAVE * 4, RING_OFF = V_OFF + 3 * V_FLOATS * 4,
              LDS_BYTES = RING_OFF + 8 * RD * 1024;  // 32 (two z windows per wave) + 54 + 72 KiB = 158 KiB
static_assert(LDS_BYTES <= 160 * 1024, "LDS");

typedef float f32x16 __attribute__((ext_vector_type(16)));

// B^T (6 x 6) of F(4,3): rows applied to a column of 6 values
__device__ __forceinline__ void bt6(const float (&d)[6], float (&o)[6]) {
  const float t1 = fmaf(d[2], -4.f, d[4]), t2 = fmaf(d[1], -4.f, d[3]), t3 = d[4] - d[2], t4 = d[3] - d[1];
  o[0] = fmaf(d[0], 4.f, fmaf(d[2], -5.f, d[4]));
  o[1] = t1 + t2;
  o[2] = t1 - t2;
  o[3] = fmaf(t4, 2.f, t3);
  o[4] = fmaf(t4, -2.f, t3);
  o[5] = fmaf(d[1], 4.f, fmaf(d[3], -5.f, d[5]));
}

// A^T (4 x 6) of F(4,3) on 4-cout vectors
__device__ __forceinline__ void at4(const f32x4 (&m)[6], f32x4 (&y)[4]) {
  const f32x4 s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
  y[0] = m[0] + s12 + s34;
  y[1] = d12 + 2.f * d34;
  y[2] = s12 + 4.f * s34;
  y[3] = d12 + 8.f * d34 + m[5];
}

// Position order.  The 36 positions (i, j) of the transformed tile are numbered so that the two rows a transform lane produces are 12
// consecutive numbers = three aligned groups of four: rows {1, 2}, {3, 4}, {0, 5} (the pairs share their column arithmetic: row 1 / 2 =
// u +- v etc.), xi = 12 * pair + 6 * (second row of the pair) + j.  Weights (pack kernel), V, accumulators and epilogue all use xi.
__host__ __device__ constexpr int pos_row(int xi) { return xi < 12 ? 1 + xi / 6 : xi < 24 ? 3 + (xi - 12) / 6 : xi < 30 ? 0 : 5; }
__host__ __device__ constexpr int pos_col(int xi) { return xi % 6; }
__host__ __device__ constexpr int pos_of(int i, int j) { return (i == 1 ? 0 : i == 2 ? 6 : i == 3 ? 12 : i == 4 ? 18 : i == 0 ? 24 : 30) + j; }
__device__ __forceinline__ bool inner_pos(int xi) {   // positions (i, j) with 1 <= i, j <= 4: where a centre-tap (1x1) kernel lives
  const int i = pos_row(xi), j = pos_col(xi);
  return i >= 1 && i <= 4 && j >= 1 && j <= 4;
}

// LDS-DMA (as in conv_wino4.hip): each lane's 16 bytes come from (wave-uniform base + per-lane byte offset) and land at LDS byte
// (M0 + lane * 16).  EVERY vector-memory load of this kernel's loop is one of these, so the s_waitcnt vmcnt(N) below are exact counts
// (vector-memory operations of a wave complete in order).
// a pointer that is wave-uniform by construction, said so where the compiler's divergence analysis cannot see it ("s" asm operands)
template <typename T>
__device__ __forceinline__ const T* uniform_ptr(const T* q) {
  const unsigned long long ub = reinterpret_cast<unsigned long long>(q);
  return reinterpret_cast<const T*>(((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(ub >> 32)) << 32) |
                                    (unsigned)__builtin_amdgcn_readfirstlane((unsigned)ub));
}
__device__ __forceinline__ void glds16(const void* sbase_, unsigned voff, unsigned lds_dst) {
  const char* sbase = uniform_ptr(reinterpret_cast<const char*>(sbase_));
  lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
template <int N>
__device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Schedule.  On gfx950 the float32 MFMA runs on the vector ALU's own multipliers: v_mfma_f32_16x16x4_f32 (32 cycles) and v_fma_f32 never
// overlap, neither inside a wave nor between the two waves of a SIMD -- another wave's vector instructions next to MFMAs take LONGER than
// the two one after the other (scripts/mfma_valu_coissue.hip, profiles/r06_mfma_valu_coissue.txt) -- so a chunk costs its 2 x 72 MFMAs per
// SIMD plus the producer's vector instructions whatever the schedule, and the schedule is the simple one: all eight waves run
//     the MFMAs of chunk c on V[c % 3]   |   the producer step of chunk c + 2 into V[(c + 2) % 3]   |   one barrier
// (three V buffers: chunk c + 1 was finished before the previous barrier, nobody reads V[(c + 2) % 3] any more).  The producer is
// wave-private: wave w owns the two tiles (row w >> 1, columns 2 (w & 1), + 1), requests their 6 x 10-pixel window by DMA into one of its
// two z windows two chunks ahead (activations come from HBM), converts it in place and transforms it -- no barrier inside.  Variants measured
// and not adopted (profiles/r06_wino44f.txt): the two halves of the workgroup half a chunk apart (one multiplies while its SIMD partner
// produces), and the producer step woven into the MFMA steps (scripts/conv_wino44f_fused.patch): same ticks.
template <bool ACT, bool SKIP>
__global__ __launch_bounds__(NTH, 2) void conv_wino44f_kernel(ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* const vbuf0 = reinterpret_cast<float*>(smem + V_OFF);
  FD_T2(const unsigned long long t2_entry = __builtin_amdgcn_s_memtime(); unsigned long long t2_mfma = 0, t2_prod = 0, t2_pwait = 0, t2_pwork = 0;)

  // ---- tile decode with XCD-aware remap (as conv_mfma.hip); the cout blocks of a pixel tile are neighbours (they share the halo in L2)
  const int bid = blockIdx.x, nblk = gridDim.x;
  int lid;
  {
    const int xcd = bid & 7, qq = nblk >> 3, rr = nblk & 7;
    lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  }
  const int cb = lid % p.tiles_n;
  int pt = lid / p.tiles_n;
  const int tw_i = pt % p.tiles_w; pt /= p.tiles_w;
  const int th_i = pt % p.tiles_h;
  const int b = pt / p.tiles_h;
  const int h0 = th_i * TH, w0 = tw_i * TW;
  const int H = p.H, W = p.W;
  const size_t img_elems = (size_t)H * W;

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int n16 = lane & 15, kq = lane >> 4;

  // ---- chunk list: the 3x3 segments first, the folded-shortcut segments (taps == 1) behind them (wave-uniform scalar code)
  int n3 = 0, n1 = 0;
  for (int s = 0; s < p.nseg; ++s) (p.seg[s].taps == 9 ? n3 : n1) += p.seg[s].C / CK;
  const int nchunk = n3 + n1;
  // A cursor walks the chunk list (segment by segment, CK channels at a time) in scalar registers: the kernel arguments of a segment are
  // read when the cursor enters it, never per chunk.  Past the last chunk it stays there (a request then re-reads that chunk: harmless).
  struct Cursor { int s, left; const float* src; int C, c0, aoff; };
  auto cur_enter = [&](Cursor& q, int sidx) {
    q.s = sidx; q.C = p.seg[sidx].C; q.aoff = p.seg[sidx].aff_off; q.c0 = 0; q.left = q.C / CK - 1;
    q.src = reinterpret_cast<const float*>(p.seg[sidx].src) + (size_t)b * img_elems * q.C;
  };
  auto cur_next = [&](Cursor& q) {
    if (q.left > 0) { --q.left; q.c0 += CK; }
    else if (q.s + 1 < p.nseg) cur_enter(q, q.s + 1);
  };
  Cursor qd, qc;        // the chunk the next halo request fetches / the chunk the next producer step converts
  cur_enter(qd, 0);
  cur_enter(qc, 0);
  int kd = 0;           // chunks requested so far (window kd & 1)

  // ---- this wave's halo window: tiles (ty, 2 tp), (ty, 2 tp + 1); lane l < 60 owns pixel (l / 10, l % 10): both 16-byte channel halves
  // (the half is uniform per request and per conversion pass: the GroupNorm affine stays in scalar registers)
  const int ty_w = wave >> 1, tp_w = wave & 1;
  float* const zw = reinterpret_cast<float*>(smem + Z_OFF) + wave * 2 * ZWAVE;   // two windows: chunk k in window k & 1
  const bool hreal = lane < WPIX;
  int hpix;
  bool hok;
  {
    const int hr = hreal ? lane / WC : 0, hc = hreal ? lane - hr * WC : 0;
    const int gh = h0 - 1 + 4 * ty_w + hr, gw = w0 - 1 + 8 * tp_w + hc;
    hok = hreal && gh >= 0 && gh < H && gw >= 0 && gw < W;
    hpix = hok ? gh * W + gw : 0;
  }
  const unsigned zlds = (unsigned)(Z_OFF + wave * 2 * ZWAVE * 4);
  auto dma_halo = [&]() {              // the next chunk -> z window kd & 1 (always 2 operations: the counted waits rely on it)
    if (hreal) {                       // (lanes 60 .. 63 masked off: their slots belong to the other half)
#pragma unroll
      for (int i = 0; i < 2; ++i) glds16(qd.src, (unsigned)((hpix * qd.C + qd.c0) * 4 + i * 16), zlds + (unsigned)((kd & 1) * ZWAVE * 4 + i * ZHALF * 4));
    }
    ++kd;
    cur_next(qd);
  };
  // the GroupNorm affine (a, d) of the 8 channels of chunk qc: 16 scalar registers, requested one producer step ahead (the scalar load is
  // invisible to hipcc: aff_wait re-defines the registers before their first use)
  f32x16 af;
  auto aff_request = [&]() {
    if constexpr (ACT) {
      if (qc.aoff >= 0) {
        const float* ap = uniform_ptr(p.affine + ((size_t)b * p.affC + qc.aoff + qc.c0) * 2);
        asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=&s"(af) : "s"(ap) : "memory");
      }
    }
  };
  auto aff_wait = [&]() {
    if constexpr (ACT) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(af)::"memory");
  };
  auto convert_halo = [&](int c) {     // own slots: [silu(a x + d)], zeros outside the image (AFTER the activation); in place
    const bool act = ACT && qc.aoff >= 0;
    aff_wait();
    if (!hreal || (!act && hok)) return;               // raw input inside the image: the slot is already what the transform reads
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float* zp = zw + (c & 1) * ZWAVE + i * ZHALF + lane * 4;
      f32x4 v = *reinterpret_cast<const f32x4*>(zp);
      if (act) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fd_silu(fmaf(v[e], af[8 * i + 2 * e], af[8 * i + 2 * e + 1]));
      }
      if (!hok) v = f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(zp) = v;
    }
  };

  // ---- 2-D input transform of the wave's two tiles: lanes 0 .. 47 = (channel 8, tile 2, third 3); a lane computes the two rows of its
  // third's pair (rows {1, 2} / {3, 4} / {0, 5}) of B^T d for the six columns, transforms them along the row and stores its positions
  // 12 third .. 12 third + 11 (see "Position order") as three 16-byte groups of four consecutive positions: V[g][tile][slot = ch ^ (tile >> 1)][4] (the XOR keeps the MFMA's ds_read_b128 free of
  // bank conflicts).  LDS operations of one wave complete in order: the reads see the in-place conversion of the other lanes.
  const int tr_ch = lane & 7, tr_tl = (lane >> 3) & 1, tr_th = lane >> 4;
  // per-lane coefficients of the column pass: the lane's two rows a, b of B^T d as  u = d4 + k1 d2,  v = k2 d3 + k3 d1,
  //   a = u + ta v + e0 d0,  b = sb u + tb v + e5 d5      (rows 1, 2: u +- v with (k1, k2, k3) = (-4, 1, -4); rows 3, 4: (-1, 2, -2);
  //   rows 0, 5: a = 4 d0 - 5 d2 + d4, b = 4 d1 - 5 d3 + d5 with (-5, -5, 4))  -- 8 operations per column instead of the full B^T (14) + selects
  const float ck1 = tr_th == 0 ? -4.f : tr_th == 1 ? -1.f : -5.f, ck2 = tr_th == 0 ? 1.f : tr_th == 1 ? 2.f : -5.f,
              ck3 = tr_th == 0 ? -4.f : tr_th == 1 ? -2.f : 4.f, cta = tr_th == 2 ? 0.f : 1.f, ce0 = tr_th == 2 ? 4.f : 0.f,
              csb = tr_th == 2 ? 0.f : 1.f, ctb = tr_th == 2 ? 1.f : -1.f, ce5 = tr_th == 2 ? 1.f : 0.f;
  auto transform = [&](int k) {        // (all 48 lanes run the same instructions: the third only sets coefficients and the store address)
    const int buf = k % 3;
    const float* zp = zw + (k & 1) * ZWAVE + (tr_ch >> 2) * ZHALF + (4 * tr_tl) * 4 + (tr_ch & 3);
    float dd[6][6];     // all 36 values requested at once: one LDS latency per producer step, not one per column
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
      for (int r = 0; r < 6; ++r) dd[j][r] = zp[(r * WC + j) * 4];
    __builtin_amdgcn_sched_barrier(0);
    float trow[2][6];   // trow[ii][j] = (B^T d)[row ii of the lane's pair][j]
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const float u = fmaf(ck1, dd[j][2], dd[j][4]);
      const float v = fmaf(ck2, dd[j][3], ck3 * dd[j][1]);
      trow[0][j] = fmaf(ce0, dd[j][0], fmaf(cta, v, u));
      trow[1][j] = fmaf(ce5, dd[j][5], fmaf(ctb, v, csb * u));
    }
    const int tile = ty_w * 4 + 2 * tp_w + tr_tl;
    float* vb = vbuf0 + buf * V_FLOATS + (tile * CK + (tr_ch ^ (tile >> 1))) * 4 + 3 * tr_th * 16 * CK * 4;
    float lin[12];
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
      float o6[6];
      bt6(trow[ii], o6);
#pragma unroll
      for (int j = 0; j < 6; ++j) lin[6 * ii + j] = o6[j];
    }
#pragma unroll
    for (int g = 0; g < 3; ++g) *reinterpret_cast<f32x4*>(vb + g * 16 * CK * 4) = f32x4{lin[4 * g], lin[4 * g + 1], lin[4 * g + 2], lin[4 * g + 3]};
  };
  // PRODUCE chunk k (this wave's two tiles) into V[k % 3], then request chunk k + 2 into the window just consumed (the request flies for a
  // whole period: activations come from HBM).  FIRST: prologue form of the wait; steady state: younger than the request of chunk k are the
  // 2 operations of the request of chunk k + 1 and of the weight pieces at most the RD of the ring.
  auto produce = [&](int k, auto first_tag) {
    FD_T2(const unsigned long long tp0 = __builtin_amdgcn_s_memtime();)
    if constexpr (decltype(first_tag)::value) vm_wait<0>(); else vm_wait<RD + 2>();
    FD_T2(const unsigned long long tp1 = __builtin_amdgcn_s_memtime(); t2_pwait += tp1 - tp0;)
    if (k < nchunk) {
      convert_halo(k);
      if (tr_th < 3) transform(k);
      cur_next(qc);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the transform's reads of z are done before the next request overwrites it
    dma_halo();             // chunk k + 2
    if (k + 1 < nchunk) aff_request();
    FD_T2(t2_pwork += __builtin_amdgcn_s_memtime() - tp1;)
  };
  using TFIRST = std::integral_constant<bool, true>;
  using TSTEADY = std::integral_constant<bool, false>;

  // ---- weights: [cout block][chunk][wave][18 = group x channel quad][64 lanes][4 positions] f32: 1 KiB per piece, straight into this
  // wave's ring of RD slots by DMA; piece s (counted over all chunks) lives in slot s % RD = (s % 18) % RD
  const char* wgt = reinterpret_cast<const char*>(reinterpret_cast<const float*>(p.w) + ((size_t)cb * (nchunk + 1) * 8 + wave) * (WSTEP * 256));
  asm volatile("" : "+s"(wgt));
  const size_t wchunk = (size_t)8 * WSTEP * 1024;   // bytes per chunk (all 8 waves)
  const unsigned lane16 = (unsigned)(lane * 16);
  const unsigned ring = (unsigned)(RING_OFF + wave * RD * 1024);
  const float* const ringp = reinterpret_cast<const float*>(smem + RING_OFF) + wave * RD * 256 + lane * 4;
  // piece idx + RD relative to the current chunk (the buffer is padded by one zero chunk: the stream runs RD pieces past the last one)
  auto dma_w = [&](int idx) { glds16(idx < WSTEP ? wgt + idx * 1024 : wgt + wchunk + (idx - WSTEP) * 1024, lane16, ring + (unsigned)((idx % RD) * 1024)); };

  f32x4 acc[36];
#pragma unroll
  for (int i = 0; i < 36; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- prologue: the first ring of weights, chunks 0 and 1 through the producer
#pragma unroll
  for (int i = 0; i < RD; ++i) dma_w(i);
  dma_halo();
  dma_halo();
  aff_request();
  produce(0, TFIRST{});
  produce(1, TFIRST{});
  __syncthreads();
  FD_T2(const unsigned long long t2_first = __builtin_amdgcn_s_memtime();)

  // one chunk of MFMAs on V[buf]; ALL = every position (3x3 chunk) or the 16 inner ones (shortcut chunk).  Step idx = (g, kk): A operand =
  // 16 bytes of ring slot idx % RD (4 positions of this lane's cout / channel), B operand = 16 bytes of V (lane = tile n16, channel
  // 4 kk + kq); both for step idx + 1 are requested before the MFMAs of step idx.  In flight on the vector-memory counter when step idx
  // waits for piece idx + 1, oldest first: the pieces idx + 1 .. idx + RD - 1 and -- for the pieces requested before this phase, idx + 1 < RD
  // -- the 2 halo requests of the producer step in front of this phase: vmcnt(RD - 2 + 2) resp. vmcnt(RD - 2).
  auto mfma_chunk = [&](int buf, auto all_tag) {
    constexpr bool ALL = decltype(all_tag)::value;
    const float* vb = vbuf0 + buf * V_FLOATS + n16 * CK * 4;
    auto b_of = [&](int idx) { const int g = idx >> 1, kk = idx & 1; return *reinterpret_cast<const f32x4*>(vb + g * 16 * CK * 4 + (((4 * kk + kq) ^ (n16 >> 1)) * 4)); };
    auto a_of = [&](int idx) { return *reinterpret_cast<const f32x4*>(ringp + (idx % RD) * 256); };
    vm_wait<RD + 1>();      // piece 0: pieces 1 .. RD - 1 and the 2 halo requests may stay in flight
    f32x4 av = a_of(0), bv = b_of(0);
#pragma unroll
    for (int idx = 0; idx < WSTEP; ++idx) {
      const int g = idx >> 1;
      f32x4 an = av, bn = bv;
      if (idx + 1 < WSTEP) {
        if (idx + 1 < RD) vm_wait<RD>(); else vm_wait<RD - 2>();
        an = a_of(idx + 1);
        bn = b_of(idx + 1);
      }
      __builtin_amdgcn_sched_barrier(0);   // the requests of step idx + 1 stay IN FRONT of the MFMAs of step idx (a whole step of LDS latency hidden)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (ALL || inner_pos(4 * g + e)) acc[4 * g + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[e], acc[4 * g + e], 0, 0, 0);
      dma_w(idx + RD);      // into the slot of step idx: its operand was read one step ago
      av = an; bv = bn;
    }
    wgt += wchunk;
    asm volatile("" : "+s"(wgt));
  };
  using TALL = std::integral_constant<bool, true>;
  using TINNER = std::integral_constant<bool, false>;

  // ---- K loop: the MFMAs of chunk c, the producer step of chunk c + 2, one barrier.  TWO loops, one per chunk kind: with both MFMA bodies in
  // one loop hipcc keeps a second copy of the 16 accumulator tiles the shortcut body touches and moves 64 registers per iteration
  auto k_iteration = [&](int c, auto all_tag) {
    FD_T2(const unsigned long long ta = __builtin_amdgcn_s_memtime();)
    mfma_chunk(c % 3, all_tag);
    FD_T2(const unsigned long long tb = __builtin_amdgcn_s_memtime(); t2_mfma += tb - ta;)
    produce(c + 2, TSTEADY{});
    __syncthreads();
    FD_T2(t2_prod += __builtin_amdgcn_s_memtime() - tb;)
  };
  for (int c = 0; c < n3; ++c) k_iteration(c, TALL{});
  for (int c = n3; c < nchunk; ++c) k_iteration(c, TINNER{});
  vm_wait<0>();                    // the weight stream's run-off and the last halo request land before the workgroup may retire
  FD_T2(const unsigned long long t2_loop = __builtin_amdgcn_s_memtime();)

  // ---- epilogue.  acc[6 i + j][r]: cout cb * 128 + 16 wave + 4 kq + r, tile n16 = (ty, tx), position (i, j).  Per output row a: T_j =
  // sum_i A^T[a][i] M[i][j] (six 4-cout vectors), Y[a][b] = sum_j A^T[b][j] T_j -> pixel (4 ty + a, 4 tx + b), 16-byte stores.
  const int cout = cb * BN + 16 * wave + 4 * kq;
  float* const out = reinterpret_cast<float*>(p.out) + (size_t)b * img_elems * p.Cout + cout;
  const float* const skip = SKIP ? reinterpret_cast<const float*>(p.skip) + (size_t)b * img_elems * p.Cout + cout : nullptr;
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) bias4 = *reinterpret_cast<const f32x4*>(p.bias + (size_t)(p.bias_rows > 1 ? b : 0) * p.Cout + cout);
  const int ty = n16 >> 2, tx = n16 & 3;
  f32x4 ssum = {0.f, 0.f, 0.f, 0.f}, ssq = {0.f, 0.f, 0.f, 0.f};
  auto row_off = [&](int a) { return ((size_t)(h0 + 4 * ty + a) * W + w0 + 4 * tx) * p.Cout; };
  f32x4 sk[2][4];   // the residual input of output row a is requested while row a - 1 is transformed
  if constexpr (SKIP) {
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) sk[0][bb] = *reinterpret_cast<const f32x4*>(skip + row_off(0) + (size_t)bb * p.Cout);
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    if constexpr (SKIP) {
      if (a < 3) {
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) sk[(a + 1) & 1][bb] = *reinterpret_cast<const f32x4*>(skip + row_off(a + 1) + (size_t)bb * p.Cout);
      }
    }
    f32x4 tj[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const f32x4 m0 = acc[pos_of(0, j)], m1 = acc[pos_of(1, j)], m2 = acc[pos_of(2, j)], m3 = acc[pos_of(3, j)], m4 = acc[pos_of(4, j)], m5 = acc[pos_of(5, j)];
      const f32x4 s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
      tj[j] = a == 0 ? m0 + s12 + s34 : a == 1 ? d12 + 2.f * d34 : a == 2 ? s12 + 4.f * s34 : d12 + 8.f * d34 + m5;
    }
    f32x4 y[4];
    at4(tj, y);
    const size_t rowoff = row_off(a);
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) {
#pragma clang fp contract(off)
      f32x4 v = y[bb] + bias4;
      if constexpr (SKIP) v = v + sk[a & 1][bb];
      v = v * p.scale;
      ssum += v;
      ssq += v * v;
      *reinterpret_cast<f32x4*>(out + rowoff + (size_t)bb * p.Cout) = v;
    }
  }
  if (p.stats) {
    // the 16 tiles of the workgroup's 16 x 16 pixels are the 16 lanes that share kq: fold them, lane n16 == 0 writes the tile's partial
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float s = ssum[r], q = ssq[r];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
      ssum[r] = s; ssq[r] = q;
    }
    if (n16 == 0) {
      const int tile = th_i * p.tiles_w + tw_i;
      float* st = p.stats + (((size_t)b * p.tiles_h * p.tiles_w + tile) * p.CoutPad + cout) * 2;
      *reinterpret_cast<f32x4*>(st) = f32x4{ssum[0], ssq[0], ssum[1], ssq[1]};
      *reinterpret_cast<f32x4*>(st + 4) = f32x4{ssum[2], ssq[2], ssum[3], ssq[3]};
    }
  }
  FD_T2(
  if (p.dbg && t == 0 && bid < 8192) {
    const unsigned long long t2_end = __builtin_amdgcn_s_memtime();
    unsigned long long* d = p.dbg + (size_t)bid * 8;
    d[0] = t2_first - t2_entry; d[1] = t2_loop - t2_first; d[2] = t2_end - t2_loop; d[3] = t2_mfma; d[4] = t2_prod; d[5] = t2_pwait; d[6] = t2_pwork;
  }
  if (p.dbg && t == 256 && bid < 8192) p.dbg[(size_t)bid * 8 + 7] = t2_mfma;   // a wave of group 1
  )
}

// ---- weight packing: [Cout][Cin][3][3] f32 (+ optional [Cout][S] 1x1 shortcut) -> [cout block of 128][chunk (+ 1 zero chunk)][wave 8]
// [18 = position group x channel quad][lane 64][4 positions]; U = G g G^T (3x3) or G[:,1] G[:,1]^T w (1x1: centre tap)
__device__ __forceinline__ float g_row(int i, int a) {   // G (6 x 3) of F(4,3)
  constexpr float G[6][3] = {{0.25f, 0.f, 0.f}, {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                             {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0.f, 0.f, 1.f}};
  return G[i][a];
}

__global__ void wino44f_pack_kernel(const float* __restrict__ w, const float* __restrict__ w_sc, float* __restrict__ dst, int Cout, int Cin, int S) {
  const int n3 = Cin / CK, nchunk = n3 + S / CK;
  const long long total = (long long)(Cout / BN) * (nchunk + 1) * 8 * WSTEP * 256;
  for (long long o = blockIdx.x * (long long)blockDim.x + threadIdx.x; o < total; o += (long long)gridDim.x * blockDim.x) {
    const int e = (int)(o & 3), lane = (int)((o >> 2) & 63);
    long long r = o >> 8;
    const int idx = (int)(r % WSTEP); r /= WSTEP;
    const int wave = (int)(r % 8); r /= 8;
    const int chunk = (int)(r % (nchunk + 1));
    const int cb = (int)(r / (nchunk + 1));
    const int g = idx >> 1, kk = idx & 1;
    const int xi = 4 * g + e, i = pos_row(xi), j = pos_col(xi);
    const int co = cb * BN + 16 * wave + (lane & 15), k = 4 * kk + (lane >> 4);
    float v = 0.f;
    if (chunk < n3) {
      const float* gp = w + ((size_t)co * Cin + chunk * CK + k) * 9;
      double acc = 0.0;
      for (int a = 0; a < 3; ++a)
        for (int bq = 0; bq < 3; ++bq) acc += (double)g_row(i, a) * (double)gp[a * 3 + bq] * (double)g_row(j, bq);
      v = (float)acc;
    } else if (chunk < nchunk) {
      v = (float)((double)g_row(i, 1) * (double)g_row(j, 1) * (double)w_sc[(size_t)co * S + (chunk - n3) * CK + k]);
    }
    dst[o] = v;
  }
}

template <bool ACT>
int set_attr44() {
  FD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino44f_kernel<ACT, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  FD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino44f_kernel<ACT, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  return FD_OK;
}
template <bool ACT>
void launch44(const ConvArgs& a, dim3 grid, hipStream_t st) {
  if (a.skip) hipLaunchKernelGGL((conv_wino44f_kernel<ACT, true>), grid, dim3(NTH), LDS_BYTES, st, a);
  else hipLaunchKernelGGL((conv_wino44f_kernel<ACT, false>), grid, dim3(NTH), LDS_BYTES, st, a);
}

}  // namespace

int fd_wino44f_init_attributes() {
  FD_TRY(set_attr44<false>());
  FD_TRY(set_attr44<true>());
  return FD_OK;
}

bool fd_wino44f_supported(int Cout, int C0, int C1, int S0, int S1, int ksize) {
  return ksize == 3 && Cout > 0 && Cout % BN == 0 && C0 > 0 && C0 % CK == 0 && C1 % CK == 0 && S0 % CK == 0 && S1 % CK == 0 && (S1 == 0 || S0 > 0);
}
bool fd_wino44f_shape_ok(int H, int W) { return H % TH == 0 && W % TW == 0; }

long long fd_wino44f_packed_bytes(int Cout, int C0, int C1, int S0, int S1) {
  return (long long)(Cout / BN) * ((C0 + C1 + S0 + S1) / CK + 1) * 8 * WSTEP * 256 * (long long)sizeof(float) + 4096;
}

int fd_wino44f_pack_weights(const float* w, const float* w_sc, void* packed, int Cout, int C0, int C1, int S0, int S1, hipStream_t st) {
  const long long total = (long long)(Cout / BN) * ((C0 + C1 + S0 + S1) / CK + 1) * 8 * WSTEP * 256;
  const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  hipLaunchKernelGGL(wino44f_pack_kernel, dim3(blocks), dim3(256), 0, st, w, w_sc, (float*)packed, Cout, C0 + C1, w_sc ? S0 + S1 : 0);
  FD_LAUNCH_CHECK();
  return FD_OK;
}

int fd_wino44f_launch(ConvArgs a, hipStream_t st) {
  FD_REQUIRE(fd_wino44f_shape_ok(a.H, a.W), "fd_conv2d: FD_WINOGRAD44 needs H %% 16 == 0 and W %% 16 == 0 (got %d x %d)", a.H, a.W);
  a.tiles_h = a.H / TH;
  a.tiles_w = a.W / TW;
  a.tiles_n = a.Cout / BN;
  a.CoutPad = fd_conv_cout_pad(a.Cout);
  const long long nblk = (long long)a.B * a.tiles_h * a.tiles_w * a.tiles_n;
  FD_REQUIRE(nblk > 0 && nblk < (1ll << 31), "conv grid out of range");
  const dim3 grid((unsigned)nblk);
  if (a.affine) launch44<true>(a, grid, st);
  else launch44<false>(a, grid, st);
  FD_LAUNCH_CHECK();
  return FD_OK;
}
