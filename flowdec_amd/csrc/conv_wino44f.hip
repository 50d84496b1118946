// conv_wino44f.hip -- 3x3 convolution as 2-D Winograd F(4x4, 3x3) in EXACT FLOAT32 on the gfx950 matrix cores (round 6).
//
// Same contract as the other convolution kernels (ddpm_conv3x3, flowdec/backbones/ncsnpp_utils/layers.py:128-134, with the
// ResnetBlockBigGANpp surroundings of layerspp.py:252-284 fused in: GroupNorm+SiLU operand transform, time / conv bias, folded 1x1
// shortcut conv, residual, 1/sqrt(2), statistics of the output for the next GroupNorm, virtual channel concat) for the fp32 mode:
//   V = B^T d B (6 x 6 input tile d, stride 4),  U = G g G^T,  M_(i,j) = sum_c U_(i,j) V_(i,j) (36 products per 16 outputs),  Y = A^T M A
// i.e. 2.25 multiply-adds per output and input channel instead of 9 (direct) or 4.5 (conv_wino4f.hip: F(4,3) along W only) -- a quarter /
// half of the MFMAs.  In float32 this pays in full: v_mfma_f32_16x16x4_f32 runs at the f32 VECTOR rate (1/16 of the fp16 matrix rate), so
// the launch is bound by the matrix pipe and nothing else matters much -- which is why this kernel, unlike the fp16 kernels of this
// library, is deliberately SIMPLE: no LDS-DMA rings, no counted waits, two barriers per chunk.  (In fp16 the 2-D form is neither accurate
// -- the transforms amplify an operand rounding ~100 x -- nor affordable: 2.25 accumulator planes per output.)
//
//   * one workgroup = 16 x 16 output pixels (16 tiles of 4 x 4) x 128 output channels, 8 waves; wave w owns couts 16 w .. 16 w + 15 for ALL
//     36 positions and all 16 tiles: 36 accumulator tiles of 16 x 16 (v_mfma_f32_16x16x4_f32) = 144 registers, the whole output
//     transform stays inside a lane (no exchange between waves);
//   * K walks 8-channel chunks: halo (18 x 18 x 8, global -> registers one chunk ahead) -> [silu(a x + d)] -> z (LDS) -> 2-D input
//     transform -> V (LDS, double-buffered, [position group of 4][tile][channel slot][4 positions]: one ds_read_b128 feeds 4 MFMAs);
//   * weights: transformed at pack time, laid out so that a lane's operand of 4 consecutive positions is one aligned 16-byte piece of
//     a contiguous 1-KiB wave load: they go from L2 straight into registers, 6 loads ahead (32 B / clk and CU: half the L1 rate);
//   * the folded 1x1 shortcut (Conv_2, layerspp.py:276-284) is more K chunks on the raw shortcut input: a centre-tap kernel transforms
//     to the 16 inner positions only, so a shortcut chunk costs 16 of the 36 MFMAs -- exactly the 1x1 convolution's multiply-adds;
//   * epilogue: Y = A^T M A on 4-cout vectors in registers, bias / residual / scale, statistics by cross-lane adds, 16-byte stores.
// Error against the f64 convolution: ~1e-6 (tests/test_hip_configs.py test_conv2d_winograd44_f32).
#include <type_traits>

#include "conv_common.h"

namespace {

using namespace fdconv;

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int NTH = 512, TH = 16, TW = 16, HH = TH + 2, HW = TW + 2, BN = 128, CK = 8;
constexpr int ZROW = HW * CK + 4;                    // floats per halo row: 148 (16-byte aligned; four rows apart = 16 banks: the transform's reads are 2-way conflicted at worst)
constexpr int Z_FLOATS = HH * ZROW;                  // 2664
constexpr int V_FLOATS = 9 * 16 * CK * 4;            // [group of 4 positions][tile][channel slot][4] = 4608
constexpr int NPIECE = HH * HW * 2;                  // 648 halo pieces of 4 channels per chunk
constexpr int WDEPTH = 4;                            // weight loads in flight per wave (1 KiB each; registers: 36 x 4 accumulators leave ~100)
constexpr int WSTEP = 18;                            // weight loads per chunk and wave: 9 position groups x 2 channel quads

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

__device__ __forceinline__ bool inner_pos(int xi) {   // positions (i, j) with 1 <= i, j <= 4: where a centre-tap (1x1) kernel lives
  const int i = xi / 6, j = xi - 6 * i;
  return i >= 1 && i <= 4 && j >= 1 && j <= 4;
}

template <bool ACT, bool SKIP>
__global__ __launch_bounds__(NTH, 2) void conv_wino44f_kernel(ConvArgs p) {
  __shared__ __attribute__((aligned(16))) float zbuf[Z_FLOATS];
  __shared__ __attribute__((aligned(16))) float vbuf[2][V_FLOATS];
  FD_T2(const unsigned long long t2_entry = __builtin_amdgcn_s_memtime();)

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

  // ---- chunk list: the 3x3 segments first, the folded-shortcut segments (taps == 1) behind them
  int n3 = 0, n1 = 0;
  for (int s = 0; s < p.nseg; ++s) (p.seg[s].taps == 9 ? n3 : n1) += p.seg[s].C / CK;
  const int nchunk = n3 + n1;
  auto chunk_src = [&](int c, const float*& src, int& C, int& c0, int& aff_off) {   // chunk c -> tensor, channels, first channel, affine offset
    int s = 0, cc = c;
    while (s + 1 < p.nseg && cc >= p.seg[s].C / CK) { cc -= p.seg[s].C / CK; ++s; }
    src = reinterpret_cast<const float*>(p.seg[s].src) + (size_t)b * img_elems * p.seg[s].C;
    C = p.seg[s].C; c0 = cc * CK; aff_off = p.seg[s].aff_off;
  };

  // ---- halo: piece q = 2 * pixel + half (4 channels = 16 bytes); thread t takes q = t and, for t < 136, q = t + 512
  int hpix[2], zoff[2];
  bool hok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int q = t + i * NTH;
    const int px = q < NPIECE ? q >> 1 : 0;
    const int hr = px / HW, hc = px - hr * HW;
    const int gh = h0 - 1 + hr, gw = w0 - 1 + hc;
    hok[i] = q < NPIECE && gh >= 0 && gh < H && gw >= 0 && gw < W;
    hpix[i] = hok[i] ? gh * W + gw : 0;
    zoff[i] = hr * ZROW + hc * CK + 4 * (q & 1);
  }
  const bool has2 = t + NTH < NPIECE;
  f32x4 hreg[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  f32x4 areg[2];      // (a, d) pairs of the 4 channels of BOTH pieces (t and t + 512 have the same parity: the same channel half; ACT only)
  auto load_halo = [&](int c) {        // global -> registers (chunk c; past the end: nothing)
    if (c >= nchunk) return;
    const float* src; int C, c0, aoff;
    chunk_src(c, src, C, c0, aoff);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (i == 1 && !has2) continue;
      const int ch = c0 + 4 * ((t + i * NTH) & 1);
      hreg[i] = hok[i] ? *reinterpret_cast<const f32x4*>(src + (size_t)hpix[i] * C + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
      if (ACT && aoff >= 0 && i == 0) {
        const float* ap = p.affine + ((size_t)b * p.affC + aoff + ch) * 2;
        areg[0] = *reinterpret_cast<const f32x4*>(ap);
        areg[1] = *reinterpret_cast<const f32x4*>(ap + 4);
      }
    }
  };
  auto store_halo = [&](int c) {       // registers -> [silu(a x + d)] -> z; zero padding AFTER the activation
    const float* src; int C, c0, aoff;
    chunk_src(c, src, C, c0, aoff);
    const bool act = ACT && aoff >= 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (i == 1 && !has2) continue;
      f32x4 v = hreg[i];
      if (act) {
        v[0] = fd_silu(fmaf(v[0], areg[0][0], areg[0][1]));
        v[1] = fd_silu(fmaf(v[1], areg[0][2], areg[0][3]));
        v[2] = fd_silu(fmaf(v[2], areg[1][0], areg[1][1]));
        v[3] = fd_silu(fmaf(v[3], areg[1][2], areg[1][3]));
        if (!hok[i]) v = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      *reinterpret_cast<f32x4*>(zbuf + zoff[i]) = v;
    }
  };

  // ---- 2-D input transform z -> V[buf]: threads 0 .. 255 = (channel 8, tile row 4, tile column 4, half 2); a thread transforms the
  // 6 x 6 tile of its channel (columns first), keeps the three output rows 3 half .. 3 half + 2 and stores them as 16-byte groups of
  // four consecutive positions: V[g][tile][slot = ch ^ (tile >> 1)][4] (the XOR keeps the MFMA's ds_read_b128 free of bank conflicts)
  const int tr_ch = t & 7, tr_ty = (t >> 3) & 3, tr_tx = (t >> 5) & 3;
  auto transform_half = [&](int buf, auto hf_tag) {   // hf: output rows 3 hf .. 3 hf + 2 (wave-uniform: waves 0, 1 / 2, 3)
    constexpr int HF = decltype(hf_tag)::value;
    const float* zp = zbuf + (4 * tr_ty) * ZROW + (4 * tr_tx) * CK + tr_ch;
    float trow[3][6];   // trow[ii][j] = (B^T d)[3 HF + ii][j]
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      float d[6], o6[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) d[r] = zp[r * ZROW + j * CK];
      bt6(d, o6);
#pragma unroll
      for (int ii = 0; ii < 3; ++ii) trow[ii][j] = o6[3 * HF + ii];
    }
    const int tile = tr_ty * 4 + tr_tx;
    float* vb = vbuf[buf] + (tile * CK + (tr_ch ^ (tile >> 1))) * 4;
    float lin[18];      // positions xi = 6 i + j of this half: 18 consecutive values starting at 18 HF
#pragma unroll
    for (int ii = 0; ii < 3; ++ii) {
      float o6[6];
      bt6(trow[ii], o6);
#pragma unroll
      for (int j = 0; j < 6; ++j) lin[6 * ii + j] = o6[j];
    }
    if constexpr (HF == 0) {   // xi 0 .. 17: groups 0 .. 3 whole, the first two positions of group 4
#pragma unroll
      for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(vb + g * 16 * CK * 4) = f32x4{lin[4 * g], lin[4 * g + 1], lin[4 * g + 2], lin[4 * g + 3]};
      *reinterpret_cast<f32x2*>(vb + 4 * 16 * CK * 4) = f32x2{lin[16], lin[17]};
    } else {                   // xi 18 .. 35: the last two positions of group 4, groups 5 .. 8 whole
      *reinterpret_cast<f32x2*>(vb + 4 * 16 * CK * 4 + 2) = f32x2{lin[0], lin[1]};
#pragma unroll
      for (int g = 5; g < 9; ++g) *reinterpret_cast<f32x4*>(vb + g * 16 * CK * 4) = f32x4{lin[4 * g - 18], lin[4 * g - 17], lin[4 * g - 16], lin[4 * g - 15]};
    }
  };
  auto transform = [&](int buf) {
    if (wave < 2) transform_half(buf, std::integral_constant<int, 0>{});
    else if (wave < 4) transform_half(buf, std::integral_constant<int, 1>{});
  };

  // ---- weights: [cout block][chunk][wave][18 = group x channel quad][64 lanes][4 positions] f32, 1 KiB per load
  const float* wgt = reinterpret_cast<const float*>(p.w) + ((size_t)cb * (nchunk + 1) * 8 + wave) * (WSTEP * 256) + lane * 4;
  const size_t wchunk = (size_t)8 * WSTEP * 256;   // floats per chunk (all 8 waves)
  f32x4 wq[WDEPTH];
#pragma unroll
  for (int i = 0; i < WDEPTH; ++i) wq[i] = *reinterpret_cast<const f32x4*>(wgt + i * 256);

  f32x4 acc[36];
#pragma unroll
  for (int i = 0; i < 36; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- prologue: chunk 0 through the producer, chunk 1 into the registers
  load_halo(0);
  store_halo(0);
  __syncthreads();
  transform(0);
  load_halo(1);
  __syncthreads();
  FD_T2(const unsigned long long t2_first = __builtin_amdgcn_s_memtime();)

  // one chunk of MFMAs on V[buf]; ALL = every position (3x3 chunk) or the 16 inner ones (shortcut chunk).  B operand of (g, kk): lane
  // (tile n16, channel 4 kk + kq) reads 16 bytes = 4 positions; A operand from the weight registers (loaded WDEPTH steps ahead; the
  // buffer is padded by one chunk, so the loads past the last chunk read zeros that are never used)
  auto mfma_chunk = [&](int buf, auto all_tag) {
    constexpr bool ALL = decltype(all_tag)::value;
    const float* vb = vbuf[buf] + n16 * CK * 4;
#pragma unroll
    for (int idx = 0; idx < WSTEP; ++idx) {
      const int g = idx >> 1, kk = idx & 1;
      const f32x4 av = wq[idx % WDEPTH];
      // the load WDEPTH steps ahead: this chunk's, or -- past its 18 -- the first ones of the NEXT chunk (layout [chunk][wave][18])
      wq[idx % WDEPTH] = *reinterpret_cast<const f32x4*>(idx + WDEPTH < WSTEP ? wgt + (idx + WDEPTH) * 256 : wgt + wchunk + (idx + WDEPTH - WSTEP) * 256);
      bool any = ALL;
#pragma unroll
      for (int e = 0; e < 4; ++e) any = any || inner_pos(4 * g + e);
      if (!any) continue;
      const f32x4 bv = *reinterpret_cast<const f32x4*>(vb + g * 16 * CK * 4 + (((4 * kk + kq) ^ (n16 >> 1)) * 4));
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (ALL || inner_pos(4 * g + e)) acc[4 * g + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[e], acc[4 * g + e], 0, 0, 0);
    }
    wgt += wchunk;
  };
  using TALL = std::integral_constant<bool, true>;
  using TINNER = std::integral_constant<bool, false>;

  // ---- K loop: MFMAs of chunk c on V[c & 1]; then chunk c + 1 (in registers since the previous iteration) -> z -> V[(c + 1) & 1], and the
  // halo of chunk c + 2 is requested.  Two barriers per chunk; the two waves of a SIMD overlap each other's producer with their MFMAs.
  for (int c = 0; c < nchunk; ++c) {
    if (c < n3) mfma_chunk(c & 1, TALL{});
    else mfma_chunk(c & 1, TINNER{});
    if (c + 1 < nchunk) {
      store_halo(c + 1);
      load_halo(c + 2);
      __syncthreads();
      transform((c + 1) & 1);
      __syncthreads();
    }
  }
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
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    f32x4 tj[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const f32x4 m0 = acc[j], m1 = acc[6 + j], m2 = acc[12 + j], m3 = acc[18 + j], m4 = acc[24 + j], m5 = acc[30 + j];
      const f32x4 s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
      tj[j] = a == 0 ? m0 + s12 + s34 : a == 1 ? d12 + 2.f * d34 : a == 2 ? s12 + 4.f * s34 : d12 + 8.f * d34 + m5;
    }
    f32x4 y[4];
    at4(tj, y);
    const size_t rowoff = ((size_t)(h0 + 4 * ty + a) * W + w0 + 4 * tx) * p.Cout;
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) {
#pragma clang fp contract(off)
      f32x4 v = y[bb] + bias4;
      if constexpr (SKIP) v = v + *reinterpret_cast<const f32x4*>(skip + rowoff + (size_t)bb * p.Cout);
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
    d[0] = t2_first - t2_entry; d[1] = t2_loop - t2_first; d[2] = t2_end - t2_loop;
  }
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
    const int xi = 4 * g + e, i = xi / 6, j = xi - 6 * i;
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
void launch44(const ConvArgs& a, dim3 grid, hipStream_t st) {
  if (a.skip) hipLaunchKernelGGL((conv_wino44f_kernel<ACT, true>), grid, dim3(NTH), 0, st, a);
  else hipLaunchKernelGGL((conv_wino44f_kernel<ACT, false>), grid, dim3(NTH), 0, st, a);
}

}  // namespace

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
