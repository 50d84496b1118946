// conv_mfma.hip -- implicit-GEMM 3x3 / 1x1 convolution on the gfx950 matrix cores (v2).
//
// Replaces the reference's nn.Conv2d calls (ddpm_conv3x3 / ddpm_conv1x1,
// flowdec/backbones/ncsnpp_utils/layers.py:110-134) together with what surrounds them in
// ResnetBlockBigGANpp.forward (layerspp.py:252-284): the GroupNorm+SiLU in front (:253,:274), the
// time-embedding bias (:272-273), the 1x1 shortcut conv Conv_2 (:278-279, folded in as extra K steps),
// the residual add and 1/sqrt(2) (:281-284), the channel concat of the up path (ncsnpp.py:337), and the
// statistics pass of the NEXT GroupNorm (per-tile partial sums of the output).
//
// Mapping (NHWC activations; one workgroup = one 16x16 pixel tile x BN output channels):
//   D[cout][pixel] += W[cout][k] * X[k][pixel],  k = (segment, channel chunk, tap)
//   * weights are the MFMA "A" operand, so every lane ends up with 4 consecutive couts of one pixel;
//   * K walks 64-byte channel chunks (32 bf16 / 16 f32 channels).  Per chunk the 18x18 halo tile is staged
//     once in LDS and reused by the 9 taps as shifted windows (ds_read immediates, taps fully unrolled);
//   * per (chunk, tap) step a BN x 64 B weight slab is needed; slabs stream through a ring of LDS slots by direct-to-LDS
//     DMA (global_load_lds_dwordx4, no VGPR staging) and the 9 taps of a chunk run as the groups (0,1)(2,3)(4,5)(6,7)(8)
//     with one barrier per group; the next chunk's halo is loaded to registers right after barrier 1 and converted /
//     stored one slot per group;
//   * the operand load applies silu(a*x+d) (GroupNorm folded to a per-(b,c) affine), zero padding after it;
//   * halo rows are padded to 80 B (5 x 16 B, coprime with the 16 slots of a 256 B bank row) with a 24-pixel pitch, weight
//     rows are 64 B with the 16-byte columns XOR-swizzled by (row >> 2) & 3: every ds_read_b128 lane group hits 16
//     distinct slots;
//   * epilogue: accumulators are transposed through LDS so that global traffic (skip read, output write) is
//     16 B per lane and fully coalesced; bias / skip / scale are applied there, and per-channel sum / sum of
//     squares of the f32 result are reduced per tile for the consumer GroupNorm (deterministic, no atomics).
//
// One kernel template, several configurations (all of them run the same K order per output, so the convolution result is the same
// bits whatever the workgroup shape; MEASUREMENTS.md "Direct kernel" / "Small grids" has the measurements behind each):
//   <WM, WN, MT, NT>  wave grid and MFMA tiles per wave: 8 waves x (128 px x 64 cout) = BN 256 by default, narrower ones for few output
//                     channels and for small grids (FD_TILE_*);
//   CW                chunk-resident weight ring + fragments two phases ahead, one barrier per chunk (the low-latency configurations);
//   MIXED = 1 / 2     f32 storage around bf16 operands: rounded once at the LDS store (FD_BF16_OPERANDS), or split as hi + lo with three
//                     MFMAs per product (FD_BF16X3_OPERANDS: the f32-tolerance mode on the bf16 matrix cores);
//   SH                one halo buffer instead of two: 70 KiB of LDS, two workgroups per CU (FD_TILE_DUO128; measured, not scheduled).
//   RE                "register epilogue, continuous tiles" (round 3; bf16, Cout == BN, whole tiles, no residual input): ONE workgroup per
//                     CU walks a contiguous range of pixel tiles as one uninterrupted software pipeline -- the weight stream wraps around
//                     to slab 0 and the last chunk of tile i prefetches / activates / publishes the first halo of tile i + 1, so a tile
//                     boundary has no prologue (no exposed memory round trip, no ring refill).  The epilogue never touches LDS staging:
//                     bias / scale / bf16 rounding happen on the accumulator registers, v_permlane32_swap pairs the two 4-cout halves of
//                     a pixel so that every lane stores 16 contiguous bytes, the stores are left in flight (the first barrier of the next
//                     tile waits with a COUNTED vmcnt), and the GroupNorm partial sums are reduced over the 32 pixel lanes with
//                     v_permlane16_swap + 4 DPP steps.  Same K order per output and the same (acc + bias) * scale arithmetic: the
//                     convolution result is bit-identical to the staged epilogue; the statistics differ in summation order only.
#include <string.h>

#include <mutex>
#include <type_traits>

#include "conv_common.h"

namespace {

using namespace fdconv;
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <typename T>
struct Math;
template <>
struct Math<bf16> {
  static constexpr int EPS = 8;  // elements per 16-byte slot
  __device__ static __forceinline__ void mma(f32x16& acc, const u32x4& wf, const u32x4& pf) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf), __builtin_bit_cast(bf16x8, pf), acc, 0, 0, 0);
  }
};
template <>
struct Math<float> {
  static constexpr int EPS = 4;
  __device__ static __forceinline__ void mma(f32x16& acc, const u32x4& wf, const u32x4& pf) {
    const f32x4 a = __builtin_bit_cast(f32x4, wf), b = __builtin_bit_cast(f32x4, pf);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
  }
};

// silu(a*x+d) on the EPS channels of one 16-byte slot; the (a,d) pairs are read from the LDS table two channels
// at a time ({a0,d0,a1,d1} per ds_read_b128) to keep the register footprint small
template <typename T, int EPS>
__device__ __forceinline__ u32x4 transform_slot(u32x4 raw, const char* ad) {
  u32x4 out;
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(ad + 16 * j);
      const unsigned u = raw[j];
      const float x0 = __builtin_bit_cast(float, u << 16), x1 = __builtin_bit_cast(float, u & 0xffff0000u);
      bf16x2 r = {(bf16)fd_silu(fmaf(x0, a[0], a[1])), (bf16)fd_silu(fmaf(x1, a[2], a[3]))};
      out[j] = __builtin_bit_cast(unsigned, r);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(ad + 16 * j);
      // NB: copy the vector element to a scalar first -- __builtin_bit_cast applied directly to an ext-vector element
      // lvalue reads element 0 for every index (hipcc / ROCm 7.2)
      const unsigned u0 = raw[2 * j], u1 = raw[2 * j + 1];
      const float y0 = fd_silu(fmaf(__builtin_bit_cast(float, u0), a[0], a[1]));
      const float y1 = fd_silu(fmaf(__builtin_bit_cast(float, u1), a[2], a[3]));
      out[2 * j] = __builtin_bit_cast(unsigned, y0);
      out[2 * j + 1] = __builtin_bit_cast(unsigned, y1);
    }
  }
  return out;
}

constexpr int cmax(int a, int b) { return a > b ? a : b; }

// v + (v of the lane DPP control CTRL selects): one v_add_f32_dpp
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// v_permlane16_swap_b32 a, b: rows 1 / 3 of a <-> rows 0 / 2 of b (a row = 16 lanes), from inline asm: hipcc (ROCm 7.2) miscompiles
// `r = __builtin_amdgcn_permlane16_swap(a, b); x = f(r[0]) + f(r[1])` into v_add_f32 x, a', a' (scripts/probe_lane_ops.hip shows
// both the builtin's lane pattern and the wrong sum).  The s_nop covers the VALU-write -> DPP-read hazard the compiler cannot see.
__device__ __forceinline__ void permlane16_swap(float& a, float& b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
  bf16x2 r = {(bf16)a, (bf16)b};
  return __builtin_bit_cast(unsigned, r);
}

// MIXED configurations (f32 storage, bf16 MFMA operands): the 8 channels of one LDS slot are 32 bytes of f32 in memory; they are
// activated in f32 and rounded to bf16 HERE, at the LDS store -- the only rounding of the residual stream on its way into a conv.
// low = true: the SECOND term of the two-term bf16 split y = bf16(y) + bf16(y - bf16(y)) (SPLIT configurations, see below).
template <bool ACT>
__device__ __forceinline__ u32x4 wide_slot(u32x4 lo, u32x4 hi, const char* ad, bool low = false) {
  u32x4 out;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned u0 = j < 2 ? lo[2 * j] : hi[2 * j - 4], u1 = j < 2 ? lo[2 * j + 1] : hi[2 * j - 3];
    float y0 = __builtin_bit_cast(float, u0), y1 = __builtin_bit_cast(float, u1);
    if constexpr (ACT) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(ad + 16 * j);
      y0 = fd_silu(fmaf(y0, a[0], a[1]));
      y1 = fd_silu(fmaf(y1, a[2], a[3]));
    }
    if (low) { y0 -= (float)(bf16)y0; y1 -= (float)(bf16)y1; }
    bf16x2 r = {(bf16)y0, (bf16)y1};
    out[j] = __builtin_bit_cast(unsigned, r);
  }
  return out;
}


// CW ("chunk-resident weights", the low-latency configuration): the ring holds the slabs of TWO whole chunks (18 slots), the K loop
// has ONE barrier per chunk and every slab is requested a full chunk before its first use -- on a small grid the tap-pair ring below
// makes every barrier group wait for one memory round trip (measured: 31 us for a 256 -> 256 convolution whatever the tile width).
template <int WM, int WN, int MT, int NT, bool CW = false, bool SH = false, bool RE = false>
struct Geo {
  static constexpr int NTH = 64 * WM * WN;
  static constexpr int NP = WM * MT;       // 4x8-pixel patches per tile, arranged (NP/2) x 2
  static constexpr int TH = 4 * (NP / 2), TW = 16;
  static constexpr int HH = TH + 2, HW = TW + 2;
  static constexpr int BN = WN * NT * 32;
  static constexpr int HALO_BYTES = HH * PITCH * ROWB;
  static constexpr int W_BYTES = BN * WROWB;
  static constexpr int W_LDS = (W_BYTES + 1023) / 1024 * 1024;  // LDS size of one weight buffer (DMA granularity)
  // Weight slabs live in a ring of NWBUF = 4 LDS slots and the 9 taps of a chunk are processed as the groups
  // (0,1)(2,3)(4,5)(6,7)(8) with ONE barrier per group (5 instead of 9 per chunk): the per-barrier cost (drain + skew) is
  // amortised over twice as many MFMAs per wave.  After each barrier the slots freed by the finished group are refilled by
  // DMA with the next slabs in K order.
  static constexpr int NWBUF = CW ? 18 : 4;
  static constexpr int NHB = SH ? 1 : 2;   // halo buffers (SH: one -- the configuration that fits two workgroups per CU)
  static constexpr int MAIN_BYTES = NHB * HALO_BYTES + NWBUF * W_LDS + AFF_BYTES;
  // epilogue staging: one M-tile row of the block (WM * 32 pixels) x BN floats (+16 B pad per pixel)
  static constexpr int EP_PIX = WM * 32;
  static constexpr int EP_ROWB = BN * 4 + 16;
  static constexpr int EP_BYTES = EP_PIX * EP_ROWB;
  static constexpr int OCT = BN / 8;             // 8-channel groups per pixel
  static constexpr int PPASS = NTH / OCT;        // pixels per epilogue pass
  static constexpr int NPASS = EP_PIX / PPASS;
  static constexpr int ST_BYTES = PPASS * OCT * 80;   // statistics staging: 80-byte records (see the end of the kernel)
  // RE: no epilogue staging; a second affine table (the prefetch runs one image ahead of the MFMAs at an image boundary) and the
  // cross-wave scratch of the statistics, both in LDS that neither the halo stores nor the weight DMA ever write
  static constexpr int SCR_BYTES = WM * BN * 8;     // [WM][BN][{sum, sumsq}] f32
  static constexpr int BIAS_BYTES = 2 * BN * 4;     // [image parity][BN] f32 (the epilogue reads its bias through LDS: a global load
                                                    // there would order itself behind the output stores still in flight)
  static constexpr int LDS_BYTES = RE ? MAIN_BYTES + AFF_BYTES + SCR_BYTES + BIAS_BYTES
                                      : cmax(MAIN_BYTES, cmax(2 * EP_BYTES, ST_BYTES));   // epilogue staging is double-buffered
  static constexpr int DMA_PER_WAVE = (W_LDS / 1024 + NTH / 64 - 1) / (NTH / 64);  // DMA instructions per wave per slab
  static constexpr int PPP = NTH / 4;            // rows covered per loader pass (4 slots per row)
  static constexpr int HITER = (HH * HW + PPP - 1) / PPP;
  static_assert(EP_PIX % PPASS == 0, "epilogue pass geometry");
  static constexpr int HPG = (HITER + 2) / 3;    // tap-pair schedule: halo slots converted before each of the barriers 3, 5, 7
  static_assert(HITER <= 6, "the halo is converted in three groups of at most two slots");
};

// LDS-DMA of one 1-KiB piece from inline asm (CW configuration): invisible to hipcc's waitcnt pass, which otherwise puts a
// vmcnt(0) in front of the first ds_read after every DMA; the barriers wait for it explicitly (block_sync).
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// MIXED: 0 = storage type == operand type; 1 = f32 storage, bf16 operands; 2 (SPLIT) = f32 storage, every operand as the two-term bf16
// split x = hi + lo and every product as hi*hi + hi*lo + lo*hi (three bf16 MFMAs, f32 accumulation: 16 mantissa bits per operand,
// ~1e-5 per product instead of bf16's 4e-3 -- the f32-tolerance mode at 3x the bf16 MFMA work instead of the f32 MFMA's 16x).  A K
// step is then 16 channels: LDS row = [hi k0-7 | hi k8-15 | lo k0-7 | lo k8-15], so the two "k-halves" of the pipeline below are
// the hi and the lo fragments of the step.
template <typename T, int WM, int WN, int MT, int NT, bool SKIP, bool CW = false, int MIXED = 0, bool SH = false, bool RE = false>
__global__ __launch_bounds__(64 * WM * WN, 2) void conv_mfma_kernel(ConvArgs p) {
  using G = Geo<WM, WN, MT, NT, CW, SH, RE>;
  static_assert(!SH || (!CW && MIXED == 0), "single-halo configuration: plain storage, tap-pair ring");
  static_assert(!RE || (!SKIP && !CW && !SH && MIXED == 0 && sizeof(T) == 2), "register-epilogue configuration: bf16, tap-pair ring, no residual input");
  using TS = std::conditional_t<MIXED != 0, float, T>;   // storage type of activations / residual / output (T: MFMA operand type)
  constexpr bool SPLIT = MIXED == 2;
  static_assert(!MIXED || sizeof(T) == 2, "MIXED = f32 storage around bf16 MFMA operands");
  static_assert(!(SPLIT && CW), "no chunk-ring SPLIT configuration");
  constexpr int EPS = Math<T>::EPS;
  constexpr int CK = SPLIT ? 16 : 4 * EPS;   // channels per K chunk (one 64-byte LDS row)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const hbuf = smem;
  char* const wbuf = smem + G::NHB * G::HALO_BYTES;
  char* const afftab = wbuf + G::NWBUF * G::W_LDS;

  FD_T2(const unsigned long long t2_entry = __builtin_amdgcn_s_memtime();)
  // ---- tile decode with XCD-aware remap: consecutive logical tiles share an XCD's L2 ------------------------
  const int bid = blockIdx.x, nblk = gridDim.x;
  int lid;
  {
    const int xcd = bid & 7, qq = nblk >> 3, rr = nblk & 7;
    lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  }
  const int H = p.H, W = p.W;
  // RE: this workgroup owns a contiguous range of `tiles_left` tiles of the (b, tile row, tile column) order (tiles_n == 1).  Only a
  // countdown and the (b, row, column) counters of the compute tile and of the loader's tile stay live in scalar registers across the K
  // loop; everything a tile boundary needs besides (image size, tile counts, output / statistics pointers) is re-read there from the
  // kernel-argument segment (`KP()`): kept live, those values pushed the K loop into scalar-register spills (+7 % cycles, measured).
  int tiles_left = 1;
  int b, th_i, tw_i, h0, w0;   // the tile the MFMAs / the epilogue work on
  const int n0 = RE ? 0 : (lid % p.tiles_n) * G::BN;
  {
    int tile0 = lid;
    if constexpr (RE) {
      const long long ntl = (long long)p.B * p.tiles_h * p.tiles_w;
      tile0 = (int)(ntl * lid / nblk);
      tiles_left = (int)(ntl * (lid + 1) / nblk) - tile0;
    }
    int pt = RE ? tile0 : tile0 / p.tiles_n;
    tw_i = pt % p.tiles_w; pt /= p.tiles_w;
    th_i = pt % p.tiles_h;
    b = pt / p.tiles_h;
  }
  h0 = th_i * G::TH; w0 = tw_i * G::TW;
  // the kernel arguments through an opaque pointer: loads through it are issued where they are written, not hoisted to the kernel entry
  auto KP = [&]() {
    const __attribute__((address_space(4))) ConvArgs* k = (const __attribute__((address_space(4))) ConvArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(k));
    return k;
  };
  auto next_tile = [&](int& b_, int& th_, int& tw_, int tiles_w, int tiles_h) {   // (b, row, column) of the following tile
    if (++tw_ == tiles_w) { tw_ = 0; if (++th_ == tiles_h) { th_ = 0; ++b_; } }
  };

  const int t = threadIdx.x;
  const int q = t & 3;        // 16-byte slot inside the 64-byte chunk row
  const int prow = t >> 2;

  // ---- loader bookkeeping.  Loads are unconditional (clamped addresses); validity is applied when the
  // registers are written to LDS. ----------------------------------------------------------------------------
  int pixl[G::HITER];   // clamped pixel index inside the loader's image
  int hlds[G::HITER];   // LDS byte offset of the slot
  unsigned pvalid = 0, hexist = 0;
  int lb = b, lth = th_i, ltw = tw_i;   // tile the LOADER works on (RE: one chunk ahead of the MFMAs, i.e. possibly already the next tile)
  auto set_loader_tile = [&](int lh0, int lw0, int H, int W) {
    pvalid = 0; hexist = 0;
#pragma unroll
    for (int i = 0; i < G::HITER; ++i) {
      const int hp = prow + i * G::PPP;
      const int hpc = hp < G::HH * G::HW ? hp : 0;
      const int hr = hpc / G::HW, hc = hpc - hr * G::HW;
      const int gh = lh0 - 1 + hr, gw = lw0 - 1 + hc;
      const bool ok = gh >= 0 && gh < H && gw >= 0 && gw < W;
      pixl[i] = ok ? gh * W + gw : 0;
      hlds[i] = (hr * PITCH + hc) * ROWB + q * 16;
      if (hp < G::HH * G::HW) { hexist |= 1u << i; if (ok) pvalid |= 1u << i; }
    }
  };
  set_loader_tile(h0, w0, H, W);
  const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)p.w_bytes, 0x00020000);
  const size_t img_elems = (size_t)H * W;

  u32x4 hreg[G::HITER];
  u32x4 hreg_hi[MIXED ? G::HITER : 1];   // MIXED: channels 4..7 of the slot (f32 in memory)

  // state of the chunk being prefetched (all wave-uniform)
  __amdgpu_buffer_rsrc_t nsrd = wsrd;
  int nC = 0, nc = 0, naff = -1;
  bool nchan_ok = false;
  auto next_chunk = [&](int s, int ch) {
    const Seg sg = p.seg[s];
    const TS* src = reinterpret_cast<const TS*>(sg.src) + (size_t)lb * img_elems * sg.C;
    nsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<TS*>(src), 0, (int)(unsigned)(img_elems * sg.C * sizeof(TS)), 0x00020000);   // num_records: unsigned 32 bits
    nC = sg.C;
    const int c = SPLIT ? ch * CK + (q & 1) * 8 : ch * CK + q * EPS;   // SPLIT: slots 0,1 = hi, 2,3 = lo of the same 16 channels
    nchan_ok = c < sg.C;
    nc = nchan_ok ? c : 0;
    naff = sg.aff_off >= 0 ? (sg.aff_off + nc) * 8 + (RE ? (lb & 1) * AFF_BYTES : 0) : -1;  // byte offset of this slot's (a,d) pairs in the LDS table (RE: one table per image parity)
  };
  int npix_on = 1;   // 0: the prefetch target is unused (last chunk of the K loop) -> every lane re-reads pixel 0 (one cache line, no HBM traffic)
  auto load_halo_slot = [&](int i) {
    // byte offset inside ONE image, unsigned 32 bits: an f32 image of a 30 s clip is 2.97 GB (the buffer resource's range is 4 GiB)
    const unsigned off = ((unsigned)(pixl[i] * npix_on) * (unsigned)nC + (unsigned)nc) * (unsigned)sizeof(TS);
    hreg[i] = __builtin_amdgcn_raw_buffer_load_b128(nsrd, (int)off, 0, 0);
    if constexpr (MIXED) hreg_hi[i] = __builtin_amdgcn_raw_buffer_load_b128(nsrd, (int)(off + 16u), 0, 0);
  };
  // the LDS image of slot i, in place: silu(a*x+d) (or the storage -> operand conversion), zero padding AFTER the activation
  auto convert_slot = [&](int i) {
    u32x4 v = hreg[i];
    if constexpr (MIXED) v = naff >= 0 ? wide_slot<true>(hreg[i], hreg_hi[i], afftab + naff, SPLIT && q >= 2) : wide_slot<false>(hreg[i], hreg_hi[i], afftab, SPLIT && q >= 2);
    else if (naff >= 0) v = transform_slot<T, EPS>(v, afftab + naff);
    if (!(nchan_ok && ((pvalid >> i) & 1u))) v = u32x4{0u, 0u, 0u, 0u};
    hreg[i] = v;
  };
  auto put_slot = [&](int i, int buf) {
    if ((hexist >> i) & 1u) *reinterpret_cast<u32x4*>(hbuf + buf * G::HALO_BYTES + hlds[i]) = hreg[i];
  };
  auto store_halo_slot = [&](int i, int buf) { convert_slot(i); put_slot(i, buf); };
  // Weight slab of one step: BN rows x 80 B, contiguous in global memory with exactly the LDS image -> copied by
  // direct-to-LDS DMA (global_load_lds_dwordx4: 64 lanes x 16 B = 1 KiB per wave instruction, destination =
  // wave-uniform base + lane * 16).  No VGPR staging, no ds_write.
  // CW: n consecutive slabs starting at `first` (slab i -> ring slot i % NWBUF), the 1-KiB pieces dealt round-robin to the waves
  auto dma_chunk = [&](int first, int n, int last) {
    constexpr int NPIECE = G::W_LDS / 1024, NW = G::NTH / 64;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const char* src0 = reinterpret_cast<const char*>(p.w) + (size_t)n0 * WROWB + (t & 63) * 16;
    for (int j = wv; j < n * NPIECE; j += NW) {
      const int sl = first + j / NPIECE, pce = j % NPIECE;
      if (sl > last) break;
      glds16(src0 + (size_t)sl * p.CoutPad * WROWB + pce * 1024, (unsigned)(G::NHB * G::HALO_BYTES + (sl % G::NWBUF) * G::W_LDS + pce * 1024));
    }
  };
  // ---- per-lane fragment coordinates ---------------------------------------------------------------------------
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave - wm * WN;
  const int l31 = lane & 31, lh = lane >> 5;
  int pbase[MT];  // LDS byte offset (inside a halo buffer) of this lane's pixel for tap (0,0), k-half folded in
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    const int pi = wm * MT + mi;
    const int r = 4 * (pi >> 1) + (l31 >> 3), c = 8 * (pi & 1) + (l31 & 7);
    pbase[mi] = (r * PITCH + c) * ROWB + lh * 16;
  }
  int wbase[2][NT];  // [k-half][nj]: row = cout, 16-B column (2 * ks + lh) ^ swizzle(row)
#pragma unroll
  for (int nj = 0; nj < NT; ++nj) {
    const int row = (wn * NT + nj) * 32 + l31, sw = (row >> 2) & 3;
    wbase[0][nj] = row * WROWB + ((lh ^ sw) * 16);
    wbase[1][nj] = row * WROWB + (((2 + lh) ^ sw) * 16);
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int nj = 0; nj < NT; ++nj)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][nj][e] = 0.f;

  // ---- pipeline ----------------------------------------------------------------------------------------------------
  // Steps s = (chunk, tap).  Per step every wave issues 2 x (MT*NT) MFMAs (k-halves ks = 0, 1).  The barrier sits in the
  // MIDDLE of a step and the operand fragments are software-pipelined across it, so that each wave's LDS reads overlap
  // its own MFMAs.  All global traffic is issued right after the barrier and has one full step to land (the barrier's
  // implicit vmcnt(0) is then free):
  //   phase A:  [store halo slot regs->LDS] | read frags(s, ks=1) || MFMA(s, ks=0)
  //   barrier   (w(s+1) DMA landed, halo slots published, all reads of w(s) done)
  //   phase B:  DMA w(s+2) -> buffer of w(s) | [load halo slot global->regs] | read frags(s+1, ks=0) || MFMA(s, ks=1)
  int n9 = 0, n1 = 0;
  for (int s = 0; s < p.nseg; ++s) {
    const int nch = (p.seg[s].C + CK - 1) / CK;
    if (p.seg[s].taps == 9) n9 += nch; else n1 += nch;
  }
  const int nsteps = n9 * 9 + n1;
  constexpr int CENTER = (PITCH + 1) * ROWB;

  // Barriers of the main loop wait for this wave's LDS-DMA EXPLICITLY: hipcc's own waitcnt insertion loses a pending
  // global_load_lds across the loop back-edge (it emitted a bare lgkmcnt(0) before the first barrier of the unrolled
  // body), which let other waves read a weight piece that had not landed.  All vector-memory traffic is issued right
  // after a barrier and is needed (landed + published) at the next one.
  auto block_sync = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  // RE: the first barrier after a register epilogue.  In flight, oldest first: the last weight DMAs of the previous tile (needed now),
  // then the epilogue's output stores (RE_NST per wave) and possibly one statistics store -- those may stay in flight.
  constexpr int RE_NST = MT * NT * 2;
  int pend_st = 0;   // stores of the last epilogue that the next barrier need not wait for (wave-uniform; 0 = none)
  auto block_sync_after_epilogue = [&]() {
    if (pend_st == RE_NST) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RE_NST) : "memory");
    else if (pend_st == RE_NST + 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RE_NST + 1) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    pend_st = 0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  // Workgroup barrier that orders LDS traffic only.  __syncthreads() is a fence as well: it waits for vmcnt(0), i.e. for
  // every global STORE issued so far to be acknowledged, which in the epilogue exposes a full memory round trip per
  // staging round (the output stores of the previous round) for no reason -- nothing read here depends on them.
  auto lds_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  u32x4 wfA[NT], pfA[MT], wfB[NT], pfB[MT];
  u32x4 wfC[SPLIT ? NT : 1];   // SPLIT: the next step's hi weight fragments land here while this step's are still live
  auto read_frags = [&](u32x4 (&wf)[NT], u32x4 (&pf)[MT], const char* hb, const char* wb, int off, int ks) {
#pragma unroll
    for (int nj = 0; nj < NT; ++nj) wf[nj] = *reinterpret_cast<const u32x4*>(wb + wbase[ks][nj]);
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) pf[mi] = *reinterpret_cast<const u32x4*>(hb + pbase[mi] + off + 32 * ks);
  };
  // All MFMAs of one k-half.  The caller has already issued (in program order) the non-MFMA work of the phase; the
  // instruction-group hints ask the scheduler to spread that work BETWEEN the MFMAs (a few instructions per gap run under
  // the 32-cycle MFMA issue interval) instead of as a serial block while the matrix pipe idles.
  auto mma_all = [&](const u32x4 (&wf)[NT], const u32x4 (&pf)[MT]) {
#pragma unroll
    for (int nj = 0; nj < NT; ++nj)
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) Math<T>::mma(acc[mi][nj], wf[nj], pf[mi]);
    constexpr int NM = MT * NT * (sizeof(T) == 2 ? 1 : 4);
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 VMEM read
      __builtin_amdgcn_sched_group_barrier(0x004, 4, 0);   // 4 SALU
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);   // 3 VALU
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // 1 DS write
    }
  };
  // SPLIT schedule of one step (sets: wfA = w_hi, pfA = p_hi, wfB = w_lo, pfB = p_lo, wfC = next w_hi):
  //   phase A: read p_lo(s)                              || w_hi*p_hi, w_lo*p_hi     (p_hi and w_lo are dead afterwards)
  //   phase B: read w_hi(s+1) -> wfC, p_hi(s+1), w_lo(s+1) || w_hi*p_lo             then wfA = wfC
  auto read_w = [&](u32x4 (&wf)[NT], const char* wb, int ks) {
#pragma unroll
    for (int nj = 0; nj < NT; ++nj) wf[nj] = *reinterpret_cast<const u32x4*>(wb + wbase[ks][nj]);
  };
  auto read_p = [&](u32x4 (&pf)[MT], const char* hb, int off, int ks) {
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) pf[mi] = *reinterpret_cast<const u32x4*>(hb + pbase[mi] + off + 32 * ks);
  };
  auto mma_pair = [&](const u32x4 (&w0)[NT], const u32x4 (&w1)[NT], const u32x4 (&pf)[MT]) {   // phase A: both weight halves x p_hi
#pragma unroll
    for (int nj = 0; nj < NT; ++nj)
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        Math<T>::mma(acc[mi][nj], w0[nj], pf[mi]);
        Math<T>::mma(acc[mi][nj], w1[nj], pf[mi]);
      }
#pragma unroll
    for (int k = 0; k < 2 * MT * NT; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 VMEM read
      __builtin_amdgcn_sched_group_barrier(0x004, 4, 0);   // 4 SALU
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);   // 3 VALU
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // 1 DS write
    }
  };
  auto take_next = [&]() {
#pragma unroll
    for (int nj = 0; nj < (SPLIT ? NT : 0); ++nj) wfA[nj] = wfC[nj];
  };

  int step = 0, hcur = 0;   // step = running (chunk, tap) index = index of the weight slab in K order (RE: running over ALL tiles of the
                            // workgroup -- it only names ring slots; `lstep` counts inside the tile)
  int lstep = 0;
  int fetch = 0;            // CW: next slab to DMA; slab i lives in ring slot i % NWBUF
  const int last_step = nsteps - 1;
  auto slot_of = [&](int i) { return wbuf + (i % G::NWBUF) * G::W_LDS; };
  // Weight stream of the tap-pair ring (non-CW).  A slab (BN rows x 64 B, contiguous in global memory = the LDS image) is copied
  // in 1-KiB pieces by direct-to-LDS DMA; every wave issues DMA_PER_WAVE of them (surplus issues re-copy another piece; a partial
  // last piece over-reads into the next slab and lands in the padding of the 1-KiB-rounded LDS slot).  ONE wave-uniform pointer
  // walks the slabs in K order and the per-lane part of the address never changes, so a slab costs two scalar adds; the pointer
  // simply runs past the last slab (fd_conv_packed_bytes pads the buffer by NWBUF slabs: what lands is never read).
  constexpr unsigned WOFF = G::NHB * G::HALO_BYTES;   // LDS byte offset of the ring
  const char* const wfetch0 = reinterpret_cast<const char*>(p.w) + (size_t)n0 * WROWB;
  const char* wfetch = wfetch0;
  const size_t slab_stride = (size_t)p.CoutPad * WROWB;
  int dma_pce[G::DMA_PER_WAVE];        // the 1-KiB pieces of a slab this wave copies (wave-uniform)
  unsigned dma_voff[G::DMA_PER_WAVE];  // per-lane byte offset inside the slab
  {
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
#pragma unroll
    for (int j = 0; j < G::DMA_PER_WAVE; ++j) {
      dma_pce[j] = (wv + j * (G::NTH / 64)) % (G::W_LDS / 1024);
      dma_voff[j] = (unsigned)(dma_pce[j] * 1024 + (t & 63) * 16);
    }
  }
  auto fetch_to = [&](unsigned dst_off) {   // next slab in K order -> LDS byte offset dst_off
#pragma unroll
    for (int j = 0; j < G::DMA_PER_WAVE; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wfetch + dma_voff[j]),
                                       (__attribute__((address_space(3))) void*)(smem + dst_off + dma_pce[j] * 1024), 16, 0, 0);
    wfetch += slab_stride;
    asm volatile("" : "+s"(wfetch));   // keep the walk scalar and sequential (hipcc otherwise pre-computes the vector addresses of a whole chunk)
  };
  auto fetch_slabs = [&](int n) {   // CW only
    dma_chunk(fetch, n, last_step);
    fetch += n;
  };
  // Prologue: all global traffic of the first step is issued at once (first halo, weight ring, affine table) so that the
  // block pays ONE memory round trip before its first MFMA; the affine table (needed by the halo transform only) is
  // staged while the halo loads and the DMAs are in flight.
  next_chunk(0, 0);
#pragma unroll
  for (int i = 0; i < G::HITER; ++i) load_halo_slot(i);
  if constexpr (CW) fetch_slabs(G::NWBUF);
  else {
#pragma unroll
    for (int k = 0; k < G::NWBUF; ++k) fetch_to(WOFF + k * G::W_LDS);
  }
  auto load_afftab = [&](int img) {   // affine table of image `img`: [affC] x (a, d)   (RE: into the table of the image's parity)
    const float* ap = p.affine + (size_t)img * p.affC * 2;
    char* dst = afftab + (RE ? (img & 1) * AFF_BYTES : 0);
    for (int i = t; i < p.affC / 2; i += G::NTH) *reinterpret_cast<f32x4*>(dst + i * 16) = *reinterpret_cast<const f32x4*>(ap + i * 4);
  };
  char* const scratch = afftab + 2 * AFF_BYTES;                 // RE only (see Geo::LDS_BYTES)
  char* const biastab = scratch + G::SCR_BYTES;
  auto load_biastab = [&](int img) {   // RE: bias row of image `img` -> LDS table of the image's parity (zeros without a bias: out-of-range reads return 0)
    const auto k = KP();
    const __amdgpu_buffer_rsrc_t bisrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(k->bias), 0, k->bias ? (k->bias_rows > 1 ? k->B : 1) * k->Cout * 4 : 0, 0x00020000);
    for (int i = t; i < G::BN; i += G::NTH)
      reinterpret_cast<float*>(biastab)[(img & 1) * G::BN + i] =
          __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(bisrd, (n0 + i) * 4, (k->bias_rows > 1 ? img : 0) * k->Cout * 4, 0));
  };
  if constexpr (RE) load_biastab(b);
  if (p.affine) {
    load_afftab(b);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < G::HITER; ++i) store_halo_slot(i, 0);
  block_sync();
  FD_T2(const unsigned long long t2_first = __builtin_amdgcn_s_memtime();)
  read_frags(wfA, pfA, hbuf, wbuf, n9 > 0 ? 0 : CENTER, 0);  // (the k-half ks = 1 is addressed by passing hb + 32 / wb + 32)
  if constexpr (SPLIT) read_w(wfB, wbuf, 1);

  int cs = 0, cch = 0;  // (segment, chunk) cursor
  auto advance = [&](int& s_, int& ch_) {
    ++ch_;
    if (ch_ >= (p.seg[s_].C + CK - 1) / CK) { ++s_; ch_ = 0; }
  };

  // Two sequential loops (all 9-tap chunks, then all 1-tap shortcut chunks) so that each loop has a single MFMA
  // site pair: with both tap counts inside one loop the compiler keeps two copies of the 128-register accumulator.
  // The phases are straight-line code: work past the end of the K loop (the DMA / halo prefetch / fragment reads of the
  // last steps) is not branched around but redirected to harmless targets (re-load of the last slab / the current chunk).
  // RE: the loader moves on to the workgroup's NEXT tile while the MFMAs work on the last chunk of this one
  auto loader_to_next_tile = [&]() {
    const auto k = KP();
    const int ob = lb;
    next_tile(lb, lth, ltw, k->tiles_w, k->tiles_h);
    if (lb != ob && k->affine) load_afftab(lb);   // into the other parity's table; published by the barriers before its first use (tap 3)
    set_loader_tile(lth * G::TH, ltw * G::TW, k->H, k->W);
    cs = 0; cch = 0;
    next_chunk(0, 0);
  };
  FD_T2(unsigned long long re_t0 = RE ? __builtin_amdgcn_s_memtime() : 0ull, re_loop = 0, re_epi = 0, re_stat = 0, re_tiles = 0;)
  for (;;) {   // tile loop: a single pass unless RE
  if constexpr (CW) {
    // Low-latency K loop.  With few MFMAs per phase the two-set pipeline above is a pure latency chain (the MFMAs of phase p + 1
    // wait for reads issued one MFMA pair earlier: ~350 cycles per phase whatever the tile width, measured); here the fragments
    // run TWO phases ahead through three register sets (18 phases per chunk: the rotation is the same in every chunk), all slabs
    // of a chunk are resident, and the one barrier per chunk sits at the end of tap 7: by then every fragment of the chunk is in
    // registers (taps 8's were read during tap 7), so the barrier publishes the next halo / the next chunk's slabs AND frees this
    // chunk's buffers for the prefetch of chunk i + 2.
    u32x4 wfS[3][NT], pfS[3][MT];
#pragma unroll
    for (int nj = 0; nj < NT; ++nj) wfS[0][nj] = wfA[nj];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) pfS[0][mi] = pfA[mi];
    if (n9 > 0) read_frags(wfS[1], pfS[1], hbuf, wbuf, 0, 1);
    for (int i = 0; i < n9; ++i) {
      const bool last_chunk = (i == n9 - 1) && n1 == 0;
      if (!last_chunk) { advance(cs, cch); next_chunk(cs, cch); }
      else npix_on = 0;
      const char* hb = hbuf + hcur * G::HALO_BYTES;
      const char* hbn = hbuf + (hcur ^ 1) * G::HALO_BYTES;
      const char* wcur = wbuf + (i & 1) * 9 * G::W_LDS;
      const char* wnext = wbuf + ((i + 1) & 1) * 9 * G::W_LDS;
      const int first_off_next = (i == n9 - 1) ? CENTER : 0;
#pragma unroll
      for (int ph = 0; ph < 18; ++ph) {
        const int tap = ph >> 1;
        if (ph == 2) {
#pragma unroll
          for (int k = 0; k < G::HITER; ++k) load_halo_slot(k);
        }
        if ((ph & 1) == 0 && tap >= 4 && tap <= 6) {
#pragma unroll
          for (int k = 0; k < G::HPG; ++k)
            if ((tap - 4) * G::HPG + k < G::HITER) store_halo_slot((tap - 4) * G::HPG + k, hcur ^ 1);
        }
        const int f = ph + 2;   // fragment read in this phase: (tap f / 2, k-half f % 2) of this chunk, or tap 0 of the next one
        if (f < 18) read_frags(wfS[f % 3], pfS[f % 3], hb, wcur + (f >> 1) * G::W_LDS, (((f >> 1) / 3) * PITCH + ((f >> 1) % 3)) * ROWB, f & 1);
        else read_frags(wfS[f % 3], pfS[f % 3], hbn, wnext, first_off_next, f & 1);
        // (no scheduling hints here: the phase boundary below keeps hipcc from sinking the reads towards their first use,
        // which would turn the two-phase distance back into a wait per MFMA)
#pragma unroll
        for (int nj = 0; nj < NT; ++nj)
#pragma unroll
          for (int mi = 0; mi < MT; ++mi) Math<T>::mma(acc[mi][nj], wfS[ph % 3][nj], pfS[ph % 3][mi]);
        __builtin_amdgcn_sched_barrier(0);
        if (ph == 15) {
          block_sync();
          fetch_slabs(9);
        }
      }
      step += 9;
      hcur ^= 1;
    }
#pragma unroll
    for (int nj = 0; nj < NT; ++nj) wfA[nj] = wfS[0][nj];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) pfA[mi] = pfS[0][mi];
  } else
  for (int i = 0; i < n9; ++i) {
    const bool last_chunk = (i == n9 - 1) && n1 == 0;
    if (!last_chunk) { advance(cs, cch); next_chunk(cs, cch); }
    else if (RE && tiles_left > 1) loader_to_next_tile();             // the first halo of the next tile
    else npix_on = 0;                                                  // the prefetch of this chunk is unused
    const char* hb = hbuf + hcur * G::HALO_BYTES;
    const char* hbn = hbuf + (hcur ^ 1) * G::HALO_BYTES;
    const int first_off_next = (i == n9 - 1) ? CENTER : 0;
    // ring slots of this chunk, computed once: tap t reads slot wso[t & 3]; 9 = 1 (mod 4), so the pattern turns by one per chunk
    unsigned wso[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) wso[k] = WOFF + ((step + k) & 3) * G::W_LDS;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int imm = ((tap / 3) * PITCH + (tap % 3)) * ROWB;
      const int imm_next = (((tap + 1) / 3) * PITCH + ((tap + 1) % 3)) * ROWB;
      const bool barrier_here = (tap & 1) || tap == 8;   // last tap of its group
      const char* wb = smem + wso[tap & 3];
      const char* wbn = smem + wso[(tap + 1) & 3];
      // ---- phase A: [store halo slot] | read frags(s, ks=1) || MFMA(s, ks=0)
      // Every wait of this loop is a full drain (the barriers' explicit vmcnt(0), and hipcc's own wait before the first
      // use of a halo register is a vmcnt(0) as well as soon as LDS-DMAs are in flight), so the schedule keeps the
      // YOUNGEST vector-memory instruction at every wait point about two taps old: all halo loads are issued together
      // with the weight DMAs right after barrier 1 and converted / stored one group of slots at a time in the phases
      // that end with barriers 3, 5 and 7 (published long before the first read in phase B of tap 8).
      if (tap == 3 || tap == 5 || tap == 7) {
#pragma unroll
        for (int k = 0; k < G::HPG; ++k)
          if ((tap - 3) / 2 * G::HPG + k < G::HITER) {
            if constexpr (SH) convert_slot((tap - 3) / 2 * G::HPG + k);   // single halo buffer: kept in registers until it is free
            else store_halo_slot((tap - 3) / 2 * G::HPG + k, hcur ^ 1);
          }
      }
      if constexpr (SPLIT) { read_p(pfB, hb, imm, 1); mma_pair(wfA, wfB, pfA); }
      else { read_frags(wfB, pfB, hb, wb, imm, 1); mma_all(wfA, pfA); }
      if (barrier_here) {
        // everything issued after the previous barrier has landed and is published; all reads of the finished group's
        // slabs are complete, so their ring slots are refilled with the next slabs in K order
        if (RE && tap == 1) { if (pend_st) block_sync_after_epilogue(); else block_sync(); }
        else block_sync();
        // the slabs of steps s + 3 (and s + 4): steps 0..3 were fetched by the prologue, every barrier moves on by its group size
        if (tap == 8) fetch_to(wso[0]);
        else { fetch_to(wso[(tap + 3) & 3]); fetch_to(wso[(tap + 4) & 3]); }
        if constexpr (SH) {
          if (tap == 8) {   // every wave has read its last fragment of this chunk (block_sync drains the LDS reads): the buffer is free
#pragma unroll
            for (int k = 0; k < G::HITER; ++k) put_slot(k, 0);
          }
        }
      }
      // ---- phase B: [load halo slot] | read frags(s+1, ks=0) || MFMA(s, ks=1)
      if (tap == 1) {
#pragma unroll
        for (int k = 0; k < G::HITER; ++k) load_halo_slot(k);
      }
      if constexpr (SPLIT) {
        read_w(wfC, wbn, 0);
        if (tap < 8) read_p(pfA, hb, imm_next, 0);
        else read_p(pfA, hbn, first_off_next, 0);
        read_w(wfB, wbn, 1);
        mma_all(wfA, pfB);
        take_next();
      } else if constexpr (SH) {
        if (tap < 8) read_frags(wfA, pfA, hb, wbn, imm_next, 0);
        mma_all(wfB, pfB);
        if (tap == 8) {   // the next halo is published; its first fragments are read behind the barrier (one exposed LDS round trip per chunk)
          lds_barrier();
          read_frags(wfA, pfA, hb, wbn, first_off_next, 0);
        }
      } else {
        if (tap < 8) read_frags(wfA, pfA, hb, wbn, imm_next, 0);
        else read_frags(wfA, pfA, hbn, wbn, first_off_next, 0);
        mma_all(wfB, pfB);
      }
      ++step;
    }
    if constexpr (!SH) hcur ^= 1;
  }
  for (int i = 0; i < n1; ++i) {
    const bool m1 = (RE ? n9 * 9 + i : step) + 1 < nsteps;
    const char* hb = hbuf + hcur * G::HALO_BYTES;
    const char* hbn = hbuf + (hcur ^ 1) * G::HALO_BYTES;
    const char* wb = slot_of(step);
    const char* wbn = slot_of(step + 1);
    if (m1) { advance(cs, cch); next_chunk(cs, cch); }
    else if (RE && tiles_left > 1) loader_to_next_tile();
    else npix_on = 0;
    // the next 1-tap chunk's halo: loaded and published within this step
#pragma unroll
    for (int k = 0; k < G::HITER; ++k) load_halo_slot(k);
    if constexpr (SPLIT) { read_p(pfB, hb, CENTER, 1); mma_pair(wfA, wfB, pfA); }
    else { read_frags(wfB, pfB, hb, wb, CENTER, 1); mma_all(wfA, pfA); }
#pragma unroll
    for (int k = 0; k < G::HITER; ++k) { if constexpr (SH) convert_slot(k); else store_halo_slot(k, hcur ^ 1); }
    block_sync();
    if constexpr (!CW) fetch_to(WOFF + (step & 3) * G::W_LDS);   // slab of step + 4 into the slot this step has just finished with
                                                                   // (CW: all shortcut slabs are already in the ring -- the launcher guarantees n1 <= NWBUF)
    if constexpr (SH) {   // the single buffer is free now: publish the next chunk's pixels, then read its fragments behind a barrier
#pragma unroll
      for (int k = 0; k < G::HITER; ++k) put_slot(k, 0);
      mma_all(wfB, pfB);
      lds_barrier();
      read_frags(wfA, pfA, hb, wbn, CENTER, 0);
      ++step;
      continue;
    }
    if constexpr (SPLIT) {
      read_w(wfC, wbn, 0);
      read_p(pfA, hbn, CENTER, 0);
      read_w(wfB, wbn, 1);
      mma_all(wfA, pfB);
      take_next();
    } else {
      read_frags(wfA, pfA, hbn, wbn, CENTER, 0);
      mma_all(wfB, pfB);
    }
    ++step; hcur ^= 1;
  }
  if constexpr (RE) {
    FD_T2(const unsigned long long re_t1 = __builtin_amdgcn_s_memtime();)
    // ---- register epilogue (see the file header).  Lane (l31, lh) of wave (wm, wn) holds, for M-tile mi and N-tile nj, the
    // couts 8 qd + 4 lh + e (qd, e = 0..3) of pixel l31 of patch wm * MT + mi.
    const auto kp = KP();
    const int H_ = kp->H, W_ = kp->W, Cout_ = kp->Cout;
    const float scale_ = kp->scale;
    bf16* const outp = reinterpret_cast<bf16*>(kp->out);
    float* const stats_ = kp->stats;
    const float* const bt = reinterpret_cast<const float*>(biastab) + (b & 1) * G::BN;
    float* const scr = reinterpret_cast<float*>(scratch);
    const int row4 = lane >> 4;   // DPP row: rows 0, 1 = lh 0, rows 2, 3 = lh 1
    const bool want_stats = stats_ != nullptr;
    const f32x2 sc2 = {scale_, scale_};
#pragma unroll
    for (int nj = 0; nj < NT; ++nj) {
      const int cw = (wn * NT + nj) * 32;   // first cout of this MFMA tile inside the workgroup's BN
#pragma unroll
      for (int qp = 0; qp < 2; ++qp) {      // couts 16 qp .. 16 qp + 15 of the MFMA tile: qd = 2 qp (x[0..3]) and 2 qp + 1 (x[4..7])
        // two channels per instruction (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: the same IEEE operations as the staged epilogue's)
        f32x2 bv[4], ssum[4], ssq[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(bt + cw + 16 * qp + 8 * h + 4 * lh);
          bv[2 * h] = f32x2{b4[0], b4[1]}; bv[2 * h + 1] = f32x2{b4[2], b4[3]};
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) ssum[k] = ssq[k] = f32x2{0.f, 0.f};
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
          const int pi = wm * MT + mi;
          const int gh = h0 + 4 * (pi >> 1) + (l31 >> 3), gw = w0 + 8 * (pi & 1) + (l31 & 7);
          bf16* const op = outp + (((size_t)b * H_ + gh) * W_ + gw) * Cout_ + n0 + cw + 8 * lh;
          unsigned pk[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
#pragma clang fp contract(off)
            f32x2 x = {acc[mi][nj][8 * qp + 2 * k], acc[mi][nj][8 * qp + 2 * k + 1]};
            x = (x + bv[k]) * sc2;                                   // two roundings, then bf16: the staged epilogue's arithmetic
            ssum[k] += x;
            ssq[k] = __builtin_elementwise_fma(x, x, ssq[k]);
            pk[k] = pack_bf16x2(x[0], x[1]);
          }
          // v_permlane32_swap(X, Y): lanes 0-31 end with {X own, X of lane + 32}, lanes 32-63 with {Y of lane - 32, Y own}: the lh = 0
          // lane of a pixel gets couts 0..7 of qd = 2 qp, its lh = 1 partner couts 0..7 of qd = 2 qp + 1 -- 16 contiguous bytes each
          const auto s0 = __builtin_amdgcn_permlane32_swap(pk[0], pk[2], false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(pk[1], pk[3], false, false);
          *reinterpret_cast<u32x4*>(op + 16 * qp) = u32x4{s0[0], s1[0], s0[1], s1[1]};
        }
        if (want_stats) {
          // sum over the 32 pixel lanes of each half-wave.  v_permlane16_swap(A, B) + add folds rows {0, 1} and {2, 3}: rows 0 / 2 then
          // carry A (qd = 2 qp), rows 1 / 3 carry B (qd = 2 qp + 1); four DPP steps finish the 16 lanes of a row.
          float* const sp = scr + (size_t)(wm * G::BN + cw + 16 * qp + 8 * (row4 & 1) + 4 * (row4 >> 1)) * 2;
#pragma unroll
          for (int which = 0; which < 2; ++which)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float A = which ? ssq[e >> 1][e & 1] : ssum[e >> 1][e & 1], B = which ? ssq[2 + (e >> 1)][e & 1] : ssum[2 + (e >> 1)][e & 1];
              permlane16_swap(A, B);
              float r = A + B;
              r = dpp_add<0xB1>(r);    // quad_perm [1,0,3,2]
              r = dpp_add<0x4E>(r);    // quad_perm [2,3,0,1]
              r = dpp_add<0x141>(r);   // row_half_mirror
              r = dpp_add<0x140>(r);   // row_mirror
              if ((lane & 15) == 0) sp[e * 2 + which] = r;
            }
        }
        __builtin_amdgcn_sched_barrier(0);   // one (nj, qp) group at a time: interleaving them costs more registers than the file has
      }
    }
    pend_st = RE_NST;
    FD_T2(const unsigned long long re_t2 = __builtin_amdgcn_s_memtime();)
    if (want_stats) {
      lds_barrier();
      // (through a buffer resource: uniform base + tile offset in scalar registers, the per-thread part is 4 * o -- a 64-bit per-thread
      // address would be hoisted out of the tile loop and spilled, and its reload would drain the stores just issued)
      const int ntile = kp->tiles_h * kp->tiles_w, cpad = kp->CoutPad;
      const __amdgpu_buffer_rsrc_t stsrd = __builtin_amdgcn_make_buffer_rsrc(stats_, 0, (int)((size_t)kp->B * ntile * cpad * 8), 0x00020000);
      const int rec = ((b * ntile + th_i * kp->tiles_w + tw_i) * cpad + n0) * 2;   // first float of this tile's record (launcher: fits 31 bits in bytes)
      for (int o = t; o < 2 * G::BN; o += G::NTH) {   // o = 2 * channel + which
        float a = 0.f;
#pragma unroll
        for (int w_ = 0; w_ < WM; ++w_) a += scr[w_ * 2 * G::BN + o];
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, a), stsrd, o * 4, rec * 4, 0);
      }
      if (__builtin_amdgcn_readfirstlane(t) < 2 * G::BN) pend_st = RE_NST + 1;
    }
    FD_T2(
    {   // per-workgroup sums over its tiles: K loop | register epilogue (math + stores + lane reduction) | statistics barrier + combine
      const unsigned long long re_t3 = __builtin_amdgcn_s_memtime();
      re_loop += re_t1 - re_t0; re_epi += re_t2 - re_t1; re_stat += re_t3 - re_t2; ++re_tiles;
      if (tiles_left <= 1 && p.dbg && t == 0 && bid < 8192) {
        unsigned long long* d = p.dbg + (size_t)bid * 8;
        d[0] = t2_first - t2_entry; d[1] = re_loop; d[2] = re_epi; d[3] = re_stat; d[4] = re_tiles; d[5] = re_t3 - t2_entry;
      }
    }
    )
    if (--tiles_left <= 0) break;
    {   // on to the next tile: its first halo is published, its first weight slabs are in the ring
      const int ob = b;
      next_tile(b, th_i, tw_i, kp->tiles_w, kp->tiles_h);
      if (b != ob) load_biastab(b);   // (other parity; read by the NEXT epilogue, many barriers from here)
      h0 = th_i * G::TH; w0 = tw_i * G::TW;
    }
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int nj = 0; nj < NT; ++nj)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mi][nj][e] = 0.f;
    // the weight stream has run 4 slabs past this tile's last step, into the copies of slabs 0..3 that the packing keeps there:
    // it continues with slab 4 of the next tile (no per-slab wrap-around test in the K loop)
    wfetch = wfetch0 + 4 * slab_stride;
    asm volatile("" : "+s"(wfetch));
    read_frags(wfA, pfA, hbuf + hcur * G::HALO_BYTES, smem + WOFF + (step & 3) * G::W_LDS, 0, 0);
    FD_T2(
    re_t0 = __builtin_amdgcn_s_memtime();   // (transition -- tile decode, accumulator reset -- is counted with the total only)
    )
    continue;
  } else {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // all fragment reads / DMA done before the epilogue reuses the LDS

  FD_T2(const unsigned long long t2_loop = __builtin_amdgcn_s_memtime();)
  // ---- epilogue ------------------------------------------------------------------------------------------------------
  // MT rounds; in round mi every wave stages its acc[mi][*] (32 pixels x NT*32 couts, f32) to LDS as
  // [pixel (wm*32 + l31)][cout], then all threads sweep the WM*32 pixels with 8 couts (16/32 B) per lane.
  TS* out = reinterpret_cast<TS*>(p.out);
  const TS* skip = SKIP ? reinterpret_cast<const TS*>(p.skip) : nullptr;   // residual input (compile-time: see the sweep below)
  const int oct = t % G::OCT, prow_e = t / G::OCT;
  const int n_e = n0 + oct * 8;
  const bool n_ok = n_e < p.Cout;
  const int n_cnt = n_ok ? ((p.Cout - n_e) >= 8 ? 8 : (p.Cout - n_e)) : 0;  // 8, or 4 for the pyramid heads
  float bv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bv[j] = 0.f;
  if (p.bias && n_ok) {
    const float* bp = p.bias + (size_t)(p.bias_rows > 1 ? b : 0) * p.Cout + n_e;
#pragma unroll
    for (int j = 0; j < 8; ++j) if (j < n_cnt) bv[j] = bp[j];
  }
  float ssum[8], ssq[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) ssum[j] = ssq[j] = 0.f;
  const bool interior = h0 + G::TH <= H && w0 + G::TW <= W && n0 + G::BN <= p.Cout;   // workgroup-uniform

  FD_T2(unsigned long long t2_e1 = 0, t2_e2 = 0, t2_e3 = 0;)
  // MT rounds, staging double-buffered (one barrier per round: the writes of round r+1 go to the other buffer, and
  // every wave has finished reading round r-1 from it before it arrived at barrier r).  The skip values of a round are
  // prefetched into registers BEFORE the staging of that round so that their latency hides behind the LDS round trip.
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    char* const stage = smem + (mi & 1) * G::EP_BYTES;
    // (a) addresses + skip prefetch for this round's passes
    size_t oaddr[G::NPASS];
    bool ovalid[G::NPASS];
    u32x4 skraw[G::NPASS][sizeof(TS) == 2 ? 1 : 2];
#pragma unroll
    for (int ps = 0; ps < G::NPASS; ++ps) {
      const int pp = prow_e + ps * G::PPASS;          // staged pixel: wm' = pp >> 5, l31' = pp & 31
      const int pi = (pp >> 5) * MT + mi;
      const int gh = h0 + 4 * (pi >> 1) + ((pp & 31) >> 3), gw = w0 + 8 * (pi & 1) + (pp & 7);
      ovalid[ps] = n_ok && gh < H && gw < W;
      oaddr[ps] = ovalid[ps] ? (((size_t)b * H + gh) * W + gw) * p.Cout + n_e : (size_t)0;
      if constexpr (SKIP) {
        if (n_cnt == 8) {
          const u32x4* sp = reinterpret_cast<const u32x4*>(skip + oaddr[ps]);
          skraw[ps][0] = sp[0];
          if constexpr (sizeof(TS) == 4) skraw[ps][1] = sp[1];
        } else if (ovalid[ps]) {   // 4 channels (pyramid heads): 8 / 16 bytes
          if constexpr (sizeof(TS) == 2) { const uint2 q2 = *reinterpret_cast<const uint2*>(skip + oaddr[ps]); skraw[ps][0] = u32x4{q2.x, q2.y, 0u, 0u}; }
          else skraw[ps][0] = *reinterpret_cast<const u32x4*>(skip + oaddr[ps]);
        }
      }
    }
    // (b) stage this wave's acc[mi][*] as [pixel][cout] f32
    {
      char* dst = stage + (wm * 32 + l31) * G::EP_ROWB + (wn * NT * 32 + 4 * lh) * 4;
#pragma unroll
      for (int nj = 0; nj < NT; ++nj)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          f32x4 v = {acc[mi][nj][4 * qd], acc[mi][nj][4 * qd + 1], acc[mi][nj][4 * qd + 2], acc[mi][nj][4 * qd + 3]};
          *reinterpret_cast<f32x4*>(dst + (nj * 32 + 8 * qd) * 4) = v;
        }
    }
    lds_barrier();
    FD_T2(if (mi == 0) t2_e1 = __builtin_amdgcn_s_memtime();)
    // (c) sweep: 8 couts (16 / 32 B) per lane, fully coalesced.  All passes of the round are computed into registers
    // first and their stores are issued back to back afterwards: vmcnt counts loads AND stores, so any wait between two
    // stores (hipcc puts a vmcnt(0) in front of the first use of a residual value) would drain the previous store at full
    // memory latency.  For the same reason the residual input is a compile-time property of the kernel (SKIP): launches
    // without it have no load, hence no wait, anywhere in the epilogue.
    // The sweep is VALU-bound (~1450 vector instructions per wave and tile, of which a third were the edge handling): tiles that lie
    // inside the image with a full set of output channels -- all but the last row / column of tiles -- take a path without the
    // per-pixel validity factor, the channel-count selects and the predicated stores.
    auto sweep = [&](auto fast_tag) {
      constexpr bool FAST = decltype(fast_tag)::value;
      u32x4 packed[G::NPASS][sizeof(TS) == 2 ? 1 : 2];
#pragma unroll
      for (int ps = 0; ps < G::NPASS; ++ps) {
        const int pp = prow_e + ps * G::PPASS;
        const float* sp = reinterpret_cast<const float*>(stage + pp * G::EP_ROWB) + oct * 8;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(sp), v1 = *reinterpret_cast<const f32x4*>(sp + 4);
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        if constexpr (SKIP) {
          if constexpr (sizeof(TS) == 2) {
            const bf16x8 sk = __builtin_bit_cast(bf16x8, skraw[ps][0]);   // (the upper 4 lanes are zero in the 4-channel case)
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += (float)sk[j];
          } else {
            const f32x4 s0 = __builtin_bit_cast(f32x4, skraw[ps][0]);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += s0[j];
            if (FAST || n_cnt == 8) {
              const f32x4 s1 = __builtin_bit_cast(f32x4, skraw[ps][1]);
#pragma unroll
              for (int j = 0; j < 4; ++j) v[4 + j] += s1[j];
            }
          }
        }
        if constexpr (FAST) {
          // two channels per instruction (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: the same IEEE operations, half the issue slots).
          // The statistics take the ROUNDED product, as the edge path below does (there the validity factor stands between the
          // multiply and the add): without contract(off) hipcc fuses (v + b) * scale + sum into an fma of the unrounded product.
#pragma unroll
          for (int k = 0; k < 4; ++k) {
#pragma clang fp contract(off)
            f32x2 x = {v[2 * k], v[2 * k + 1]};
            const f32x2 b2 = {bv[2 * k], bv[2 * k + 1]}, sc2 = {p.scale, p.scale};
            f32x2 s1 = {ssum[2 * k], ssum[2 * k + 1]}, s2 = {ssq[2 * k], ssq[2 * k + 1]};
            x = (x + b2) * sc2;
            s1 += x;
            s2 = __builtin_elementwise_fma(x, x, s2);
            v[2 * k] = x[0]; v[2 * k + 1] = x[1];
            ssum[2 * k] = s1[0]; ssum[2 * k + 1] = s1[1];
            ssq[2 * k] = s2[0]; ssq[2 * k + 1] = s2[1];
          }
        } else {
          const float keep = ovalid[ps] ? 1.f : 0.f;   // pixels outside the image do not enter the statistics
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            v[j] = (v[j] + bv[j]) * p.scale;
            const float w = j < n_cnt ? v[j] * keep : 0.f;
            ssum[j] += w; ssq[j] = fmaf(w, w, ssq[j]);
          }
        }
        if constexpr (sizeof(TS) == 2) {
          bf16x8 tv;
#pragma unroll
          for (int j = 0; j < 8; ++j) tv[j] = (bf16)v[j];
          packed[ps][0] = __builtin_bit_cast(u32x4, tv);
        } else {
          packed[ps][0] = __builtin_bit_cast(u32x4, f32x4{v[0], v[1], v[2], v[3]});
          packed[ps][1] = __builtin_bit_cast(u32x4, f32x4{v[4], v[5], v[6], v[7]});
        }
      }
#pragma unroll
      for (int ps = 0; ps < G::NPASS; ++ps) {
        if (!FAST && !ovalid[ps]) continue;
        TS* op = out + oaddr[ps];
        if (FAST || n_cnt == 8) {
          *reinterpret_cast<u32x4*>(op) = packed[ps][0];
          if constexpr (sizeof(TS) == 4) *(reinterpret_cast<u32x4*>(op) + 1) = packed[ps][1];
        } else {   // 4 valid channels (pyramid heads: Cout = 4)
          if constexpr (sizeof(TS) == 2) *reinterpret_cast<uint2*>(op) = uint2{packed[ps][0][0], packed[ps][0][1]};
          else *reinterpret_cast<u32x4*>(op) = packed[ps][0];
        }
      }
    };
    if (interior) sweep(std::true_type{});
    else sweep(std::false_type{});
    FD_T2(
    if (mi == 0) t2_e2 = __builtin_amdgcn_s_memtime();
    if (mi == MT - 1) t2_e3 = __builtin_amdgcn_s_memtime();
    )
  }
  lds_barrier();

  if (p.stats) {  // per-tile partial sums of the output, reduced over the PPASS threads that share an octet
    // staging [PPASS][OCT] records of {sum, sumsq} x 8 channels = 64 B, padded to 80 B so that the four 16-byte writes of
    // the 8 lanes of a ds_write_b128 group land in distinct bank slots
    constexpr int SREC = 80;
    char* const stg = smem;
    char* rec = stg + (prow_e * G::OCT + oct) * SREC;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<f32x4*>(rec + 16 * j) = f32x4{ssum[2 * j], ssq[2 * j], ssum[2 * j + 1], ssq[2 * j + 1]};
    lds_barrier();
    const int tile = th_i * p.tiles_w + tw_i;
    for (int o = t; o < 2 * G::BN; o += G::NTH) {   // o = 2 * channel + which
      const char* src = stg + (o >> 4) * SREC + (o & 15) * 4;
      float a = 0.f;
#pragma unroll
      for (int r = 0; r < G::PPASS; ++r) a += *reinterpret_cast<const float*>(src + r * G::OCT * SREC);
      if (n0 + (o >> 1) < p.CoutPad)
        p.stats[(((size_t)b * p.tiles_h * p.tiles_w + tile) * p.CoutPad + n0) * 2 + o] = a;
    }
  }
  FD_T2(
  if (p.dbg && t == 0 && bid < 8192) {   // per-workgroup record, no atomics (they would perturb the epilogue being measured)
    const unsigned long long t2_end = __builtin_amdgcn_s_memtime();
    unsigned long long* d = p.dbg + (size_t)bid * 8;
    d[0] = t2_first - t2_entry;   // prologue: tile decode, first halo + weight ring fill, affine table
    d[1] = t2_loop - t2_first;    // main loop
    d[2] = t2_end - t2_loop;      // epilogue
    d[3] = t2_e1 - t2_loop;       // epilogue: start -> barrier of round 0 (bias load, address math, first staging)
    d[4] = t2_e2 - t2_e1;         // sweep of round 0
    d[5] = t2_e3 - t2_e2;         // rounds 1..MT-1
    d[6] = t2_end - t2_e3;        // final barrier + statistics
  }
  )
  break;
  }   // !RE
  }   // tile loop
}

// ---- weight packing: [Cout][Cin][k][k] f32 -> [step][CoutPad][WROWB bytes] -----------------------------------------
// steps enumerate (concat segment, 64-byte channel chunk, tap); inside a row the four 16-byte columns are XOR-swizzled by
// (cout >> 2) & 3 (the LDS image of a slab is a plain byte copy of this layout, see WROWB).
template <typename T>
__global__ void pack_weights_kernel(const float* __restrict__ w, char* __restrict__ dst, int Cout, int CoutPad, int C0, int C1,
                                    int taps, long long step0) {
  constexpr int CK = 64 / sizeof(T);    // channels per row
  constexpr int EPC = 16 / sizeof(T);   // elements per 16-byte column
  const int nchunk0 = (C0 + CK - 1) / CK, nchunks = nchunk0 + (C1 + CK - 1) / CK;
  const long long total = (long long)nchunks * taps * CoutPad * CK;
  const int Cin = C0 + C1;
  T* d = reinterpret_cast<T*>(dst + step0 * CoutPad * WROWB);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i % CK);
    long long r = i / CK;
    const int n = (int)(r % CoutPad); r /= CoutPad;
    const int tap = (int)(r % taps);
    const int chunk = (int)(r / taps);
    float v = 0.f;
    if (n < Cout) {
      int c;
      if (chunk < nchunk0) { c = chunk * CK + k; if (c >= C0) c = -1; }
      else { c = (chunk - nchunk0) * CK + k; c = (c < C1) ? C0 + c : -1; }
      if (c >= 0) v = w[((size_t)n * Cin + c) * taps + tap];
    }
    const int col = (k / EPC) ^ ((n >> 2) & 3);
    d[(i - k) + col * EPC + (k % EPC)] = (T)v;
  }
}

// SPLIT packing: steps of 16 channels; row = [hi k0-7 | hi k8-15 | lo k0-7 | lo k8-15] (bf16), w = hi + lo to 16 mantissa bits
__global__ void pack_weights_split_kernel(const float* __restrict__ w, char* __restrict__ dst, int Cout, int CoutPad, int C0, int C1,
                                          int taps, long long step0) {
  constexpr int CK = 16;
  const int nchunk0 = (C0 + CK - 1) / CK, nchunks = nchunk0 + (C1 + CK - 1) / CK;
  const long long total = (long long)nchunks * taps * CoutPad * CK;
  const int Cin = C0 + C1;
  bf16* d = reinterpret_cast<bf16*>(dst + step0 * CoutPad * WROWB);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i % CK);
    long long r = i / CK;
    const long long row = r;                       // (chunk * taps + tap) * CoutPad + n
    const int n = (int)(r % CoutPad); r /= CoutPad;
    const int tap = (int)(r % taps);
    const int chunk = (int)(r / taps);
    float v = 0.f;
    if (n < Cout) {
      int c;
      if (chunk < nchunk0) { c = chunk * CK + k; if (c >= C0) c = -1; }
      else { c = (chunk - nchunk0) * CK + k; c = (c < C1) ? C0 + c : -1; }
      if (c >= 0) v = w[((size_t)n * Cin + c) * taps + tap];
    }
    const bf16 hi = (bf16)v, lo = (bf16)(v - (float)hi);
    const int sw = (n >> 2) & 3;
    d[row * 32 + (((k >> 3) ^ sw) * 8) + (k & 7)] = hi;
    d[row * 32 + (((2 + (k >> 3)) ^ sw) * 8) + (k & 7)] = lo;
  }
}

inline int pad_to(int x, int a) { return (x + a - 1) / a * a; }
inline int cout_pad(int Cout) { return Cout <= 32 ? 32 : (Cout <= 128 ? 128 : pad_to(Cout, 256)); }
inline int n_steps(int C0, int C1, int taps, int CK) { return (fd_cdiv(C0, CK) + fd_cdiv(C1, CK)) * taps; }

FD_T2(
unsigned long long* g_dbg = nullptr;  // instrumented builds only: device buffer of 8 counters per workgroup (fd_debug_buffer)
)

template <typename T, int WM, int WN, int MT, int NT, bool CW = false, int MIXED = 0, bool SH = false>
int set_attr() {
  using G = Geo<WM, WN, MT, NT, CW, SH>;
  FD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_mfma_kernel<T, WM, WN, MT, NT, false, CW, MIXED, SH>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
  FD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_mfma_kernel<T, WM, WN, MT, NT, true, CW, MIXED, SH>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
  return FD_OK;
}

int g_num_cu[64] = {};   // compute units per device (fd_conv_init_attributes)

template <int WM, int WN, int MT, int NT>
int set_attr_re() {
  using G = Geo<WM, WN, MT, NT, false, false, true>;
  FD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_mfma_kernel<bf16, WM, WN, MT, NT, false, false, 0, false, true>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
  return FD_OK;
}

// RE ("register epilogue, continuous tiles"): one persistent workgroup per compute unit, each walking a contiguous range of tiles.
template <int WM, int WN, int MT, int NT>
bool re_applies(const ConvArgs& a) {
  using G = Geo<WM, WN, MT, NT, false, false, true>;
  if (a.skip || a.Cout != G::BN || a.H % G::TH || a.W % G::TW) return false;
  if ((long long)a.B * (a.H / G::TH) * (a.W / G::TW) * a.CoutPad * 8 >= (1ll << 31)) return false;   // statistics addressed through one buffer resource
  bool has9 = false;
  for (int s = 0; s < a.nseg; ++s) has9 = has9 || a.seg[s].taps == 9;
  return has9;
}
template <int WM, int WN, int MT, int NT>
int launch_conv_re(ConvArgs a, hipStream_t st) {
  using G = Geo<WM, WN, MT, NT, false, false, true>;
  a.tiles_h = a.H / G::TH; a.tiles_w = a.W / G::TW; a.tiles_n = 1;
  const long long ntl = (long long)a.B * a.tiles_h * a.tiles_w;
  FD_REQUIRE(ntl > 0 && ntl < (1ll << 31), "conv grid out of range");
  int dev = 0;
  FD_HIP(hipGetDevice(&dev));
  const int ncu = dev >= 0 && dev < 64 && g_num_cu[dev] > 0 ? g_num_cu[dev] : 256;
  const unsigned grid = (unsigned)(ntl < ncu ? ntl : ncu);
  hipLaunchKernelGGL((conv_mfma_kernel<bf16, WM, WN, MT, NT, false, false, 0, false, true>), dim3(grid), dim3(G::NTH), G::LDS_BYTES, st, a);
  FD_LAUNCH_CHECK();
  return FD_OK;
}

template <typename T, int WM, int WN, int MT, int NT, bool CW = false, int MIXED = 0, bool SH = false>
int launch_conv(ConvArgs a, hipStream_t st) {
  using G = Geo<WM, WN, MT, NT, CW, SH>;
  a.tiles_h = fd_cdiv(a.H, G::TH);
  a.tiles_w = fd_cdiv(a.W, G::TW);
  a.tiles_n = fd_cdiv(a.Cout, G::BN);
  const long long nblk = (long long)a.B * a.tiles_h * a.tiles_w * a.tiles_n;
  FD_REQUIRE(nblk > 0 && nblk < (1ll << 31), "conv grid out of range");
  if (a.skip) hipLaunchKernelGGL((conv_mfma_kernel<T, WM, WN, MT, NT, true, CW, MIXED, SH>), dim3((unsigned)nblk), dim3(G::NTH), G::LDS_BYTES, st, a);
  else hipLaunchKernelGGL((conv_mfma_kernel<T, WM, WN, MT, NT, false, CW, MIXED, SH>), dim3((unsigned)nblk), dim3(G::NTH), G::LDS_BYTES, st, a);
  FD_LAUNCH_CHECK();
  return FD_OK;
}

// bn_hint: output channels per workgroup (32 / 64 / 128 / 256) or 0 = by Cout.  All configurations run the same K order per
// output, so the convolution result does not depend on the choice (the per-tile statistics differ in summation order only).
template <typename T>
int dispatch_conv(const ConvArgs& a, hipStream_t st, int bn_hint, bool chunk_ring, bool persist = false) {
  int bn = a.Cout <= 32 ? 32 : (a.Cout <= 128 ? 128 : 256);
  if (bn_hint > 0 && bn_hint < bn) bn = bn_hint;   // (bn_hint < 0: FD_TILE_DUO128, below)
  if constexpr (sizeof(T) == 2) {
    if (bn_hint == -128 && a.Cout % 128 == 0) return launch_conv<T, 2, 2, 4, 2, false, 0, true>(a, st);   // FD_TILE_DUO128
    if (chunk_ring && bn <= 64) {   // low-latency configurations: bf16, at most 18 folded-shortcut (or 1x1) steps
      int n1 = 0;
      for (int s = 0; s < a.nseg; ++s) if (a.seg[s].taps == 1) n1 += fd_cdiv(a.seg[s].C, 32);
      if (n1 <= 18) return bn == 64 ? launch_conv<T, 4, 2, 2, 1, true>(a, st) : launch_conv<T, 4, 1, 2, 1, true>(a, st);
    }
  }
  if (bn == 32) return launch_conv<T, 4, 1, 2, 1>(a, st);                                 // 4 waves, BN = 32 (pyramid heads, tiny grids)
  if (bn == 64) return launch_conv<T, 4, 2, 2, 1>(a, st);                                 // 8 waves, BN = 64
  if constexpr (sizeof(T) == 2) {   // FD_TILE_PERSIST: whole tiles, Cout == BN, no residual input -> the persistent register-epilogue configuration
    if (persist && bn == 128 && re_applies<4, 2, 2, 2>(a)) return launch_conv_re<4, 2, 2, 2>(a, st);
    if (persist && bn == 256 && re_applies<2, 4, 4, 2>(a)) return launch_conv_re<2, 4, 4, 2>(a, st);
  }
  if (bn == 128) return launch_conv<T, 4, 2, 2, 2>(a, st);                                // 8 waves, BN = 128
  return launch_conv<T, 2, 4, 4, 2>(a, st);                                               // 8 waves, BN = 256
}

// f32 storage around bf16 MFMA operands (FD_F32 | FD_BF16_OPERANDS): the three default widths
int dispatch_conv_mixed(const ConvArgs& a, hipStream_t st) {
  if (a.Cout <= 32) return launch_conv<bf16, 4, 1, 2, 1, false, 1>(a, st);
  if (a.Cout <= 128) return launch_conv<bf16, 4, 2, 2, 2, false, 1>(a, st);
  return launch_conv<bf16, 2, 4, 4, 2, false, 1>(a, st);
}
// two-term bf16 split (FD_F32 | FD_BF16X3_OPERANDS)
int dispatch_conv_split(const ConvArgs& a, hipStream_t st) {
  if (a.Cout <= 32) return launch_conv<bf16, 4, 1, 2, 1, false, 2>(a, st);
  if (a.Cout <= 128) return launch_conv<bf16, 4, 2, 2, 2, false, 2>(a, st);
  return launch_conv<bf16, 2, 4, 4, 2, false, 2>(a, st);
}

}  // namespace

// hipFuncSetAttribute (dynamic LDS above 64 KiB) is a per-device property: done once per device, thread-safe
int fd_conv_init_attributes() {
  static std::mutex mu;
  static bool done_dev[64] = {};
  int dev = 0;
  FD_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  const bool known = dev >= 0 && dev < 64;
  if (known && done_dev[dev]) return FD_OK;
  FD_TRY((set_attr<bf16, 4, 1, 2, 1>())); FD_TRY((set_attr<bf16, 4, 2, 2, 1>())); FD_TRY((set_attr<bf16, 4, 2, 2, 1, true>())); FD_TRY((set_attr<bf16, 4, 1, 2, 1, true>())); FD_TRY((set_attr<bf16, 2, 2, 4, 2, false, 0, true>())); FD_TRY((set_attr<bf16, 4, 2, 2, 2>())); FD_TRY((set_attr<bf16, 2, 4, 4, 2>()));
  FD_TRY((set_attr<float, 4, 1, 2, 1>())); FD_TRY((set_attr<float, 4, 2, 2, 1>())); FD_TRY((set_attr<float, 4, 2, 2, 2>())); FD_TRY((set_attr<float, 2, 4, 4, 2>()));
  FD_TRY((set_attr<bf16, 4, 1, 2, 1, false, 1>())); FD_TRY((set_attr<bf16, 4, 2, 2, 2, false, 1>())); FD_TRY((set_attr<bf16, 2, 4, 4, 2, false, 1>()));
  FD_TRY((set_attr<bf16, 4, 1, 2, 1, false, 2>())); FD_TRY((set_attr<bf16, 4, 2, 2, 2, false, 2>())); FD_TRY((set_attr<bf16, 2, 4, 4, 2, false, 2>()));
  FD_TRY((set_attr_re<2, 4, 4, 2>())); FD_TRY((set_attr_re<4, 2, 2, 2>()));
  if (known) FD_HIP(hipDeviceGetAttribute(&g_num_cu[dev], hipDeviceAttributeMultiprocessorCount, dev));
  FD_TRY(fd_wino_init_attributes());
  FD_TRY(fd_wino4_init_attributes());
  FD_TRY(fd_wino4f_init_attributes());
  FD_TRY(fd_wino44f_init_attributes());
  FD_TRY(fd_head_init_attributes());
  FD_TRY(fd_headf_init_attributes());
  if (known) done_dev[dev] = true;
  return FD_OK;
}

FD_T2(extern "C" int fd_debug_buffer(void* p) { g_dbg = reinterpret_cast<unsigned long long*>(p); return FD_OK; })

extern "C" int fd_conv_cout_pad(int Cout) { return cout_pad(Cout); }
extern "C" int fd_conv_stats_tiles(int H, int W) { return fd_cdiv(H, 16) * fd_cdiv(W, 16); }

extern "C" long long fd_conv_packed_bytes(int Cout, int C0, int C1, int ksize, int S0, int S1, int wdtype) {
  if (wdtype == (FD_F32 | FD_WINOGRAD44)) return fd_wino44f_supported(Cout, C0, C1, S0, S1, ksize) ? fd_wino44f_packed_bytes(Cout, C0, C1, S0, S1) : 0;   // 2-D, exact f32
  if (wdtype == (FD_F32 | FD_WINOGRAD4)) return fd_wino4f_supported(Cout, C0, C1, S0, S1, ksize) ? fd_wino4f_packed_bytes(Cout, C0, C1, S0, S1) : 0;   // exact f32
  if (wdtype & FD_WINOGRAD4) return fd_wino4_supported(Cout, C0, C1, S0, S1, ksize) && (wdtype & 0xff) == FD_BF16 ? fd_wino4_packed_bytes(Cout, C0, C1, S0, S1) : 0;
  if (wdtype & FD_WINOGRAD) return fd_wino_supported(Cout, C0, C1, S0, S1, ksize) && (wdtype & 0xff) == FD_BF16 ? fd_wino_packed_bytes(Cout, C0, C1, S0, S1) : 0;
  if (wdtype == (FD_F32 | FD_BF16_OPERANDS)) wdtype = FD_BF16;   // weights follow the OPERAND type
  const int CK = wdtype == FD_BF16 ? 32 : 16;                    // (FD_F32 | FD_BF16X3_OPERANDS: 16 channels per step, hi + lo per row)
  FD_REQUIRE(wdtype == FD_BF16 || wdtype == FD_F32 || wdtype == (FD_F32 | FD_BF16X3_OPERANDS), "fd_conv_packed_bytes: bad dtype");
  // + 1 KiB slack: the DMA of a partial last 1-KiB piece (BN = 32 configuration) over-reads past the final slab
  // + 4 slabs: the kernel's slab pointer runs up to a ring length past the last step without a clamp (what it fetches there is never read)
  return (long long)(n_steps(C0, C1, ksize * ksize, CK) + n_steps(S0, S1, 1, CK) + 4) * cout_pad(Cout) * WROWB + 1024;
}

extern "C" int fd_conv_pack_weights(const float* w, const float* w_sc, void* packed, int Cout, int C0, int C1, int ksize, int S0,
                                    int S1, int wdtype, void* stream) {
  FD_REQUIRE(w && packed, "fd_conv_pack_weights: null pointer");
  FD_REQUIRE(ksize == 1 || ksize == 3, "fd_conv_pack_weights: ksize must be 1 or 3");
  FD_REQUIRE((S0 + S1 == 0) == (w_sc == nullptr), "fd_conv_pack_weights: shortcut weight / channel mismatch");
  if (wdtype == (FD_F32 | FD_WINOGRAD44)) {   // 2-D F(4x4, 3x3) in exact float32 (conv_wino44f.hip)
    FD_REQUIRE(fd_wino44f_supported(Cout, C0, C1, S0, S1, ksize), "fd_conv_pack_weights: FD_F32 | FD_WINOGRAD44 needs ksize 3, Cout %% 128 == 0 and channel counts %% 8 == 0");
    return fd_wino44f_pack_weights(w, w_sc, packed, Cout, C0, C1, S0, S1, fd_stream(stream));
  }
  if (wdtype == (FD_F32 | FD_WINOGRAD4)) {   // F(4,3) in exact float32 (conv_wino4f.hip)
    FD_REQUIRE(fd_wino4f_supported(Cout, C0, C1, S0, S1, ksize), "fd_conv_pack_weights: FD_F32 | FD_WINOGRAD4 needs ksize 3, Cout == 256 and channel counts %% 16 == 0");
    return fd_wino4f_pack_weights(w, w_sc, packed, Cout, C0, C1, S0, S1, fd_stream(stream));
  }
  if (wdtype & FD_WINOGRAD4) {
    FD_REQUIRE((wdtype & 0xff) == FD_BF16 && fd_wino4_supported(Cout, C0, C1, S0, S1, ksize),
               "fd_conv_pack_weights: FD_WINOGRAD4 needs bf16 storage, ksize 3, Cout == 256 and channel counts %% 32 == 0");
    return fd_wino4_pack_weights(w, w_sc, packed, Cout, C0, C1, S0, S1, fd_stream(stream));
  }
  if (wdtype & FD_WINOGRAD) {
    FD_REQUIRE((wdtype & 0xff) == FD_BF16 && fd_wino_supported(Cout, C0, C1, S0, S1, ksize),
               "fd_conv_pack_weights: FD_WINOGRAD needs bf16 storage, ksize 3, Cout %% 128 == 0 and channel counts %% 32 == 0");
    return fd_wino_pack_weights(w, w_sc, packed, Cout, C0, C1, S0, S1, fd_stream(stream));
  }
  if (wdtype == (FD_F32 | FD_BF16_OPERANDS)) wdtype = FD_BF16;   // weights follow the OPERAND type
  const bool split = wdtype == (FD_F32 | FD_BF16X3_OPERANDS);
  FD_REQUIRE(wdtype == FD_BF16 || wdtype == FD_F32 || split, "fd_conv_pack_weights: bad dtype");
  const int taps = ksize * ksize, CoutPad = cout_pad(Cout), CK = wdtype == FD_BF16 ? 32 : 16;
  hipStream_t st = fd_stream(stream);
  auto run = [&](const float* src, int c0, int c1, int tp, long long step0) {
    const long long total = (long long)n_steps(c0, c1, tp, CK) * CoutPad * CK;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    if (split)
      hipLaunchKernelGGL(pack_weights_split_kernel, dim3(blocks), dim3(256), 0, st, src, (char*)packed, Cout, CoutPad, c0, c1, tp, step0);
    else if (wdtype == FD_BF16)
      hipLaunchKernelGGL(pack_weights_kernel<bf16>, dim3(blocks), dim3(256), 0, st, src, (char*)packed, Cout, CoutPad, c0, c1, tp, step0);
    else
      hipLaunchKernelGGL(pack_weights_kernel<float>, dim3(blocks), dim3(256), 0, st, src, (char*)packed, Cout, CoutPad, c0, c1, tp, step0);
  };
  run(w, C0, C1, taps, 0);
  if (w_sc) run(w_sc, S0, S1, 1, n_steps(C0, C1, taps, CK));
  FD_LAUNCH_CHECK();
  {   // the 4 slabs of padding behind the last step hold copies of slabs 0..3: the weight stream of a persistent workgroup
      // (FD_TILE_PERSIST) runs over the end of a tile's K loop straight into the first steps of the next tile, no wrap-around test
    const long long ns = n_steps(C0, C1, taps, CK) + n_steps(S0, S1, 1, CK);
    const size_t slab = (size_t)CoutPad * WROWB;
    if (ns >= 4) FD_HIP(hipMemcpyAsync((char*)packed + ns * slab, packed, 4 * slab, hipMemcpyDeviceToDevice, st));
  }
  return FD_OK;
}

extern "C" int fd_conv2d(const void* in0, int C0, const void* in1, int C1, const float* affine, const void* sc0, int S0,
                         const void* sc1, int S1, const void* packed_w, const float* bias, int bias_rows, const void* skip,
                         float scale, void* out, int Cout, float* stats, int B, int H, int W, int ksize, int dtype, void* stream) {
  FD_REQUIRE(in0 && packed_w && out, "fd_conv2d: null pointer");
  FD_REQUIRE(ksize == 1 || ksize == 3, "fd_conv2d: ksize must be 1 or 3 (got %d)", ksize);
  const bool wino = (dtype & FD_WINOGRAD) != 0, wino4 = (dtype & FD_WINOGRAD4) != 0, wino44 = (dtype & FD_WINOGRAD44) != 0;
  FD_REQUIRE(!(wino && wino4) && !(wino44 && (wino || wino4)), "fd_conv2d: FD_WINOGRAD, FD_WINOGRAD4 and FD_WINOGRAD44 exclude each other");
  FD_REQUIRE(!wino44 || dtype == (FD_F32 | FD_WINOGRAD44), "fd_conv2d: FD_WINOGRAD44 goes with pure FD_F32 (no operand / tile flags)");
  const bool mixed = (dtype & FD_BF16_OPERANDS) != 0, split = (dtype & FD_BF16X3_OPERANDS) != 0;
  FD_REQUIRE(!(mixed || split) || dtype == (FD_F32 | FD_BF16_OPERANDS) || dtype == (FD_F32 | FD_BF16X3_OPERANDS),
             "fd_conv2d: FD_BF16_OPERANDS / FD_BF16X3_OPERANDS go with FD_F32 storage and the default direct configuration only");
  const bool reversed = (dtype & FD_TILE_REVERSED) != 0;
  FD_REQUIRE(!reversed || wino4, "fd_conv2d: FD_TILE_REVERSED goes with FD_WINOGRAD4 only");
  const int tile = dtype & FD_TILE_MASK;
  const int bn_hint = tile == FD_TILE_DUO128 ? -128 : (tile == FD_TILE_BN32 || tile == FD_TILE_BN32_CHUNK) ? 32 : (tile == FD_TILE_BN64 || tile == FD_TILE_BN64_CHUNK) ? 64 : tile == FD_TILE_BN128 ? 128 : 0;
  FD_REQUIRE(tile == 0 || bn_hint != 0 || tile == FD_TILE_PERSIST, "fd_conv2d: bad FD_TILE_* flag");
  dtype &= 0xff;
  const bool wino4f = wino4 && dtype == FD_F32 && !mixed && !split;   // F(4,3) in exact float32 (conv_wino4f.hip)
  FD_REQUIRE(!wino4f || (fd_wino4f_supported(Cout, C0, C1, S0, S1, ksize) && fd_wino4f_shape_ok(H, W)),
             "fd_conv2d: FD_F32 | FD_WINOGRAD4 needs ksize 3, Cout == 256, channel counts %% 16 == 0, H %% 16 == W %% 16 == 0");
  FD_REQUIRE(!wino4 || wino4f || (dtype == FD_BF16 && fd_wino4_supported(Cout, C0, C1, S0, S1, ksize) && fd_wino4_shape_ok(H, W)),
             "fd_conv2d: FD_WINOGRAD4 needs bf16 storage (or pure FD_F32), ksize 3, Cout == 256, channel counts %% 32 == 0, H %% 16 == W %% 16 == 0");
  FD_REQUIRE(!wino || (dtype == FD_BF16 && fd_wino_supported(Cout, C0, C1, S0, S1, ksize)),
             "fd_conv2d: FD_WINOGRAD needs bf16 storage, ksize 3, Cout %% 128 == 0 and channel counts %% 32 == 0");
  FD_REQUIRE(dtype == FD_BF16 || dtype == FD_F32, "fd_conv2d: dtype must be FD_BF16 (bf16 MFMA) or FD_F32 (f32 MFMA)");
  FD_REQUIRE(C0 > 0 && C0 % 8 == 0 && C1 >= 0 && C1 % 8 == 0, "fd_conv2d: input channels must be multiples of 8 (C0=%d C1=%d)", C0, C1);
  FD_REQUIRE(S0 >= 0 && S0 % 8 == 0 && S1 >= 0 && S1 % 8 == 0, "fd_conv2d: shortcut channels must be multiples of 8 (S0=%d S1=%d)", S0, S1);
  FD_REQUIRE((C1 == 0) == (in1 == nullptr) && (S0 == 0) == (sc0 == nullptr) && (S1 == 0) == (sc1 == nullptr),
             "fd_conv2d: tensor / channel-count mismatch");
  FD_REQUIRE(S1 == 0 || S0 > 0, "fd_conv2d: sc1 without sc0");
  FD_REQUIRE(Cout > 0 && (Cout % 8 == 0 || Cout == 4), "fd_conv2d: Cout must be 4 or a multiple of 8 (got %d)", Cout);
  FD_REQUIRE(B > 0 && H > 0 && W > 0, "fd_conv2d: bad shape");
  FD_REQUIRE(bias == nullptr || bias_rows == 1 || bias_rows == B, "fd_conv2d: bias_rows must be 1 or B");
  int cm = C0; if (C1 > cm) cm = C1; if (S0 > cm) cm = S0; if (S1 > cm) cm = S1; if (Cout > cm) cm = Cout;
  // 32-bit byte offsets inside one image: unsigned in the direct kernel (4 GiB: ~43 s of audio with f32 storage), the bf16-only kernels
  // (Winograd, heads) keep the signed range (2 GiB: ~43 s in bf16 as well)
  FD_REQUIRE((long long)H * W * cm * (dtype == FD_BF16 ? 2 : 4) < (dtype == FD_BF16 ? (1ll << 31) : (1ll << 32)),
             "fd_conv2d: one image of %d x %d x %d elements exceeds %d GiB (32-bit buffer offsets; ~43 s of audio)", H, W, cm, dtype == FD_BF16 ? 2 : 4);
  FD_TRY(fd_conv_init_attributes());
  const int taps = ksize * ksize;
  ConvArgs a{};
  int ns = 0;
  a.seg[ns++] = Seg{in0, C0, affine ? 0 : -1, taps};
  if (C1) a.seg[ns++] = Seg{in1, C1, affine ? C0 : -1, taps};
  if (S0) a.seg[ns++] = Seg{sc0, S0, -1, 1};
  if (S1) a.seg[ns++] = Seg{sc1, S1, -1, 1};
  a.nseg = ns;
  a.affine = affine; a.affC = C0 + C1;
  a.w = packed_w; a.w_bytes = fd_conv_packed_bytes(Cout, C0, C1, ksize, S0, S1, dtype | (wino ? FD_WINOGRAD : 0) | (wino4 ? FD_WINOGRAD4 : 0) | (wino44 ? FD_WINOGRAD44 : 0) | (mixed ? FD_BF16_OPERANDS : 0) | (split ? FD_BF16X3_OPERANDS : 0));
  a.bias = bias; a.bias_rows = bias_rows; a.skip = skip; a.scale = scale; a.out = out; a.Cout = Cout; a.CoutPad = cout_pad(Cout);
  a.stats = stats; a.B = B; a.H = H; a.W = W; a.reversed = reversed ? 1 : 0;
  FD_T2(a.dbg = g_dbg;)
  FD_REQUIRE(a.w_bytes < (1ll << 31), "fd_conv2d: packed weights exceed 2 GiB");
  FD_REQUIRE(affine == nullptr || (C0 + C1) * 8 <= AFF_BYTES, "fd_conv2d: at most %d activated input channels", AFF_BYTES / 8);
  if (wino44) {
    FD_REQUIRE(fd_wino44f_supported(Cout, C0, C1, S0, S1, ksize) && fd_wino44f_shape_ok(H, W),
               "fd_conv2d: FD_F32 | FD_WINOGRAD44 needs ksize 3, Cout %% 128 == 0, channel counts %% 8 == 0, H %% 16 == W %% 16 == 0");
    return fd_wino44f_launch(a, fd_stream(stream));
  }
  if (wino4f) return fd_wino4f_launch(a, fd_stream(stream));
  if (wino4) return fd_wino4_launch(a, fd_stream(stream));
  if (wino) return fd_wino_launch(a, fd_stream(stream));
  if (mixed) return dispatch_conv_mixed(a, fd_stream(stream));
  if (split) return dispatch_conv_split(a, fd_stream(stream));
  if (bn_hint == 0 && fd_head_supported(a, ksize, dtype)) return fd_head_launch(a, fd_stream(stream));
  if (bn_hint == 0 && tile == 0 && fd_headf_supported(a, ksize, dtype)) return fd_headf_launch(a, fd_stream(stream));
  if (dtype == FD_BF16) return dispatch_conv<bf16>(a, fd_stream(stream), bn_hint, tile == FD_TILE_BN64_CHUNK || tile == FD_TILE_BN32_CHUNK, tile == FD_TILE_PERSIST);
  return dispatch_conv<float>(a, fd_stream(stream), bn_hint, false);
}
