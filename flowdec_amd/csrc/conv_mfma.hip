// conv_mfma.hip -- implicit-GEMM 3x3 / 1x1 convolution on the gfx950 matrix cores.
//
// Replaces the reference's nn.Conv2d calls (ddpm_conv3x3 / ddpm_conv1x1,
// flowdec/backbones/ncsnpp_utils/layers.py:110-134) together with the GroupNorm+SiLU that
// precedes them (layerspp.py:253,274), the time-embedding bias (:272-273), the skip add and
// 1/sqrt(2) rescale (:281-284) and the channel concat of the up path (ncsnpp.py:337).
//
// Mapping (NHWC activations, one workgroup = one 16x16 pixel tile x BN output channels):
//   D[cout][pixel] += W[cout][k] * X[k][pixel],  k = (tap, cin)      (weights are the MFMA "A"
//   operand so that each lane ends up with 4 consecutive couts of ONE pixel -> 8/16-byte stores)
//   * the K loop walks 64-byte channel chunks (32 bf16 / 16 f32 channels); per chunk the (TH+2)x(TW+2)
//     halo tile is staged ONCE in LDS and re-used by all 9 taps as shifted windows;
//   * per (chunk, tap) step the BN x 64 B weight slab is staged in LDS; both are double-buffered and
//     register-prefetched one step ahead so there is a single barrier per step;
//   * the operand load applies silu(a*x+d) (GroupNorm folded to a per-(b,c) affine) and the zero
//     padding AFTER it, exactly like conv(pad(act(gn(x))));
//   * LDS rows are 64 B; the 16-byte slot index is XOR-swizzled with (row>>2)&3 and the halo pitch is
//     24 (= 8 mod 16), which makes every ds_read_b128 lane group hit 16 distinct slots (conflict-free
//     for the 4x8-pixel MFMA patches used here).
#include "common.h"
#include "internal.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
  const void* in0; const void* in1;
  int C0, C1;
  const float* affine;     // [B][C0+C1][2] or null
  const void* w;           // packed [chunk][tap][CoutPad][64 bytes]
  const float* bias;       // [bias_rows][Cout] or null
  int bias_rows;
  const void* skip;        // [B,H,W,Cout] or null
  float scale;
  void* out;
  int Cout, CoutPad;
  int B, H, W;
  int tiles_h, tiles_w, tiles_n;
  int nchunk0, nchunks;    // chunks taken from in0, total chunks
};

template <typename T>
struct Math;
template <>
struct Math<bf16> {
  static constexpr int EPS = 8;  // elements per 16-byte slot
  __device__ static void mma(f32x16& acc, const u32x4& wf, const u32x4& pf) {
    bf16x8 a = __builtin_bit_cast(bf16x8, wf), b = __builtin_bit_cast(bf16x8, pf);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  }
};
template <>
struct Math<float> {
  static constexpr int EPS = 4;
  __device__ static void mma(f32x16& acc, const u32x4& wf, const u32x4& pf) {
    f32x4 a = __builtin_bit_cast(f32x4, wf), b = __builtin_bit_cast(f32x4, pf);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
  }
};

// ad[j] holds the (a, d) pairs of channels 2j and 2j+1 of the slot: {a0, d0, a1, d1}
template <typename T, int EPS>
__device__ __forceinline__ u32x4 transform_slot(u32x4 raw, const f32x4 (&ad)[EPS / 2]) {
  if constexpr (sizeof(T) == 2) {
    bf16x8 v = __builtin_bit_cast(bf16x8, raw);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (bf16)fd_silu(fmaf((float)v[i], ad[i >> 1][2 * (i & 1)], ad[i >> 1][2 * (i & 1) + 1]));
    return __builtin_bit_cast(u32x4, v);
  } else {
    f32x4 v = __builtin_bit_cast(f32x4, raw);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = fd_silu(fmaf(v[i], ad[i >> 1][2 * (i & 1)], ad[i >> 1][2 * (i & 1) + 1]));
    return __builtin_bit_cast(u32x4, v);
  }
}

template <int TAPS, int WM, int WN, int MT, int NT>
struct Geo {
  static constexpr int NTH = 64 * WM * WN;
  static constexpr int HALO = TAPS == 9 ? 1 : 0;
  static constexpr int NP = WM * MT;       // 4x8-pixel patches per tile
  static constexpr int TH = 4 * (NP / 2);  // patches arranged (NP/2) x 2
  static constexpr int TW = 16;
  static constexpr int HH = TH + 2 * HALO, HW = TW + 2 * HALO;
  static constexpr int PITCH = 24;         // halo row pitch in pixels, = 8 (mod 16)
  static constexpr int BN = WN * NT * 32;
  static constexpr int HALO_BYTES = HH * PITCH * 64;
  static constexpr int W_BYTES = BN * 64;
  static constexpr int LDS_BYTES = 2 * HALO_BYTES + 2 * W_BYTES;
  static constexpr int PPP = NTH / 4;      // rows (pixels / couts) covered per loader pass
  static constexpr int HITER = (HH * HW + PPP - 1) / PPP;
  static constexpr int WITER = (BN + PPP - 1) / PPP;
};

template <typename T, int TAPS, int WM, int WN, int MT, int NT>
__global__ __launch_bounds__(64 * WM * WN, 2) void conv_mfma_kernel(ConvArgs p) {
  using G = Geo<TAPS, WM, WN, MT, NT>;
  constexpr int EPS = Math<T>::EPS;
  constexpr int CK = 4 * EPS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const hbuf = smem;
  char* const wbuf = smem + 2 * G::HALO_BYTES;

  // ---- tile decode with XCD-aware remap: consecutive logical tiles share an XCD's L2 ------------
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
  const int h0 = th_i * G::TH, w0 = tw_i * G::TW, n0 = nt_i * G::BN;
  const int H = p.H, W = p.W;

  const int t = threadIdx.x;
  const int q = t & 3;        // 16-byte slot inside the 64-byte chunk row
  const int prow = t >> 2;

  // ---- loader bookkeeping (independent of the chunk).  Loads are UNCONDITIONAL (addresses clamped to a
  // valid pixel / row) and validity is applied when the registers are written to LDS: conditional loads
  // make hipcc keep the staging registers in scratch and wait vmcnt(0) right after each load. -----------
  int pix[G::HITER];   // clamped global pixel index
  int hlds[G::HITER];  // LDS byte offset of the slot
  unsigned pvalid = 0, hexist = 0;
#pragma unroll
  for (int i = 0; i < G::HITER; ++i) {
    const int hp = prow + i * G::PPP;
    const int hpc = hp < G::HH * G::HW ? hp : 0;
    const int hr = hpc / G::HW, hc = hpc - hr * G::HW;
    const int gh = h0 - G::HALO + hr, gw = w0 - G::HALO + hc;
    const bool ok = gh >= 0 && gh < H && gw >= 0 && gw < W;
    pix[i] = ok ? ((b * H + gh) * W + gw) : (b * H * W);
    const int hpl = hr * G::PITCH + hc;
    hlds[i] = hpl * 64 + ((q ^ ((hpl >> 2) & 3)) << 4);
    if (hp < G::HH * G::HW) { hexist |= 1u << i; if (ok) pvalid |= 1u << i; }
  }

  u32x4 hreg[G::HITER];
  u32x4 wreg[G::WITER];
  f32x4 af[EPS / 2];
  bool chan_ok = false;

  auto load_halo = [&](int chunk) {
    const bool second = chunk >= p.nchunk0;
    const T* src = reinterpret_cast<const T*>(second ? p.in1 : p.in0);
    const int Cs = second ? p.C1 : p.C0;
    int c = (second ? chunk - p.nchunk0 : chunk) * CK + q * EPS;
    chan_ok = c < Cs;
    c = chan_ok ? c : 0;
#pragma unroll
    for (int i = 0; i < G::HITER; ++i) hreg[i] = *reinterpret_cast<const u32x4*>(src + (size_t)pix[i] * Cs + c);
    if (p.affine != nullptr) {
      const float* ap = p.affine + ((size_t)b * (p.C0 + p.C1) + (second ? p.C0 : 0) + c) * 2;
#pragma unroll
      for (int e = 0; e < EPS / 2; ++e) af[e] = *reinterpret_cast<const f32x4*>(ap + 4 * e);
    }
  };
  auto store_halo = [&](int buf) {
    char* dst = hbuf + buf * G::HALO_BYTES;
    const unsigned ok_mask = chan_ok ? pvalid : 0u;
#pragma unroll
    for (int i = 0; i < G::HITER; ++i) {
      u32x4 v = hreg[i];
      if (p.affine != nullptr) v = transform_slot<T, EPS>(v, af);
      if (!((ok_mask >> i) & 1u)) v = u32x4{0u, 0u, 0u, 0u};  // zero padding AFTER the activation
      if ((hexist >> i) & 1u) *reinterpret_cast<u32x4*>(dst + hlds[i]) = v;
    }
  };
  constexpr bool W_EXACT = (G::BN % G::PPP) == 0;
  auto load_w = [&](int step) {  // step = chunk * TAPS + tap; packed layout is [step][CoutPad][64 B]
    const char* src = reinterpret_cast<const char*>(p.w) + ((size_t)step * p.CoutPad + n0) * 64 + q * 16;
#pragma unroll
    for (int i = 0; i < G::WITER; ++i) {
      const int row = W_EXACT ? prow + i * G::PPP : (prow + i * G::PPP) % G::BN;
      wreg[i] = *reinterpret_cast<const u32x4*>(src + (size_t)row * 64);
    }
  };
  auto store_w = [&](int buf) {
    char* dst = wbuf + buf * G::W_BYTES;
#pragma unroll
    for (int i = 0; i < G::WITER; ++i) {
      const int row = prow + i * G::PPP;
      if (W_EXACT || row < G::BN) *reinterpret_cast<u32x4*>(dst + row * 64 + ((q ^ ((row >> 2) & 3)) << 4)) = wreg[i];
    }
  };

  // ---- per-lane fragment coordinates ----------------------------------------------------------------
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave - wm * WN;
  const int l31 = lane & 31, lh = lane >> 5;
  int hp_base[MT];  // halo pixel index of this lane's pixel for tap (0,0)
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    const int pi = wm * MT + mi;
    const int r = 4 * (pi >> 1) + (l31 >> 3), c = 8 * (pi & 1) + (l31 & 7);
    hp_base[mi] = r * G::PITCH + c;
  }
  int wrow_off[NT];  // byte offset of this lane's weight row, swizzle term kept separately
  int wrow_sw[NT];
#pragma unroll
  for (int nj = 0; nj < NT; ++nj) {
    const int row = (wn * NT + nj) * 32 + l31;
    wrow_off[nj] = row * 64;
    wrow_sw[nj] = (row >> 2) & 3;
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int nj = 0; nj < NT; ++nj)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][nj][e] = 0.f;

  // ---- pipeline ---------------------------------------------------------------------------------------
  const int nsteps = p.nchunks * TAPS;
  load_halo(0);
  load_w(0);
  store_halo(0);
  store_w(0);
  __syncthreads();

  int chunk = 0, tap = 0;
  for (int step = 0; step < nsteps; ++step) {
    const bool more = step + 1 < nsteps;
    const bool new_chunk = more && (tap == TAPS - 1);
    if (more) load_w(step + 1);
    if (new_chunk) load_halo(chunk + 1);

    const char* hb = hbuf + (chunk & 1) * G::HALO_BYTES;
    const char* wb = wbuf + (step & 1) * G::W_BYTES;
    const int dy = (TAPS == 9) ? tap / 3 : 0;
    const int dx = (TAPS == 9) ? tap - dy * 3 : 0;
    const int tap_off = dy * G::PITCH + dx;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int kq = 2 * ks + lh;
      u32x4 wf[NT], pf[MT];
#pragma unroll
      for (int nj = 0; nj < NT; ++nj)
        wf[nj] = *reinterpret_cast<const u32x4*>(wb + wrow_off[nj] + ((kq ^ wrow_sw[nj]) << 4));
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        const int hp = hp_base[mi] + tap_off;
        pf[mi] = *reinterpret_cast<const u32x4*>(hb + hp * 64 + ((kq ^ ((hp >> 2) & 3)) << 4));
      }
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int nj = 0; nj < NT; ++nj) Math<T>::mma(acc[mi][nj], wf[nj], pf[mi]);
    }

    if (more) store_w((step + 1) & 1);
    if (new_chunk) store_halo((chunk + 1) & 1);
    __syncthreads();
    if (++tap == TAPS) { tap = 0; ++chunk; }
  }

  // ---- epilogue: + bias, + skip, * scale, store 4 consecutive couts per lane-quad ---------------------
  T* out = reinterpret_cast<T*>(p.out);
  const T* skip = reinterpret_cast<const T*>(p.skip);
  const float* bias = p.bias ? p.bias + (size_t)(p.bias_rows > 1 ? b : 0) * p.Cout : nullptr;
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    const int pi = wm * MT + mi;
    const int gh = h0 + 4 * (pi >> 1) + (l31 >> 3), gw = w0 + 8 * (pi & 1) + (l31 & 7);
    if (gh >= H || gw >= W) continue;
    const size_t pix = ((size_t)b * H + gh) * W + gw;
#pragma unroll
    for (int nj = 0; nj < NT; ++nj) {
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int n = n0 + (wn * NT + nj) * 32 + 8 * qd + 4 * lh;
        if (n < p.Cout) {
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = acc[mi][nj][4 * qd + j];
          if (bias) {
            f32x4 bv = *reinterpret_cast<const f32x4*>(bias + n);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += bv[j];
          }
          if (skip) {
            float s[4];
            fd_load_vec<T, 4>(skip + pix * p.Cout + n, s);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += s[j];
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] *= p.scale;
          fd_store_vec<T, 4>(out + pix * p.Cout + n, v);
        }
      }
    }
  }
}

// ---- weight packing: [Cout][Cin][k][k] f32 -> [chunk][tap][CoutPad][CK] -------------------------------
template <typename T>
__global__ void pack_weights_kernel(const float* __restrict__ w, T* __restrict__ dst, int Cout, int CoutPad, int C0,
                                    int C1, int taps, int nchunk0, int nchunks) {
  constexpr int CK = 64 / sizeof(T);
  const long long total = (long long)nchunks * taps * CoutPad * CK;
  const int Cin = C0 + C1;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i % CK);
    long long r = i / CK;
    const int n = (int)(r % CoutPad); r /= CoutPad;
    const int tap = (int)(r % taps);
    const int chunk = (int)(r / taps);
    int c;  // channel index in the concatenated input, or -1 for padding
    if (chunk < nchunk0) { c = chunk * CK + k; if (c >= C0) c = -1; }
    else { c = (chunk - nchunk0) * CK + k; c = (c < C1) ? C0 + c : -1; }
    float v = 0.f;
    if (c >= 0 && n < Cout) v = w[((size_t)n * Cin + c) * taps + tap];
    dst[i] = (T)v;
  }
}

inline int pad_to(int x, int a) { return (x + a - 1) / a * a; }
// rows of the packed weight slab: a multiple of the N tile of the config that will run (32 or 128)
inline int cout_pad(int Cout) { return Cout <= 32 ? 32 : pad_to(Cout, 128); }

template <typename T, int TAPS, int WM, int WN, int MT, int NT>
int launch_conv(ConvArgs a, hipStream_t st) {
  using G = Geo<TAPS, WM, WN, MT, NT>;
  a.tiles_h = fd_cdiv(a.H, G::TH);
  a.tiles_w = fd_cdiv(a.W, G::TW);
  a.tiles_n = fd_cdiv(a.Cout, G::BN);
  auto kern = conv_mfma_kernel<T, TAPS, WM, WN, MT, NT>;
  const long long nblk = (long long)a.B * a.tiles_h * a.tiles_w * a.tiles_n;
  FD_REQUIRE(nblk > 0 && nblk < (1ll << 31), "conv grid out of range");
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(G::NTH), G::LDS_BYTES, st, a);
  FD_LAUNCH_CHECK();
  return FD_OK;
}

// > 64 KiB of dynamic LDS needs an explicit opt-in per kernel; done once (not inside a stream capture)
template <typename T, int TAPS, int WM, int WN, int MT, int NT>
int set_attr() {
  using G = Geo<TAPS, WM, WN, MT, NT>;
  FD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_mfma_kernel<T, TAPS, WM, WN, MT, NT>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
  return FD_OK;
}

template <typename T>
int dispatch_conv(ConvArgs a, int ksize, hipStream_t st) {
  constexpr int CK = 64 / sizeof(T);
  a.nchunk0 = fd_cdiv(a.C0, CK);
  a.nchunks = a.nchunk0 + fd_cdiv(a.C1, CK);
  const bool small_n = a.Cout <= 32;
  a.CoutPad = cout_pad(a.Cout);
  if (ksize == 3) {
    if (small_n) return launch_conv<T, 9, 4, 1, 2, 1>(a, st);
    return launch_conv<T, 9, 2, 2, 4, 2>(a, st);
  } else {
    if (small_n) return launch_conv<T, 1, 4, 1, 2, 1>(a, st);
    return launch_conv<T, 1, 2, 2, 4, 2>(a, st);
  }
}

}  // namespace

int fd_conv_init_attributes() {
  static bool done = false;
  if (done) return FD_OK;
  FD_TRY((set_attr<bf16, 9, 4, 1, 2, 1>())); FD_TRY((set_attr<bf16, 9, 2, 2, 4, 2>()));
  FD_TRY((set_attr<bf16, 1, 4, 1, 2, 1>())); FD_TRY((set_attr<bf16, 1, 2, 2, 4, 2>()));
  FD_TRY((set_attr<float, 9, 4, 1, 2, 1>())); FD_TRY((set_attr<float, 9, 2, 2, 4, 2>()));
  FD_TRY((set_attr<float, 1, 4, 1, 2, 1>())); FD_TRY((set_attr<float, 1, 2, 2, 4, 2>()));
  done = true;
  return FD_OK;
}

extern "C" long long fd_conv_packed_bytes(int Cout, int C0, int C1, int ksize, int wdtype) {
  const int CK = wdtype == FD_BF16 ? 32 : 16;
  return (long long)(fd_cdiv(C0, CK) + fd_cdiv(C1, CK)) * ksize * ksize * cout_pad(Cout) * 64;
}

extern "C" int fd_conv_pack_weights(const float* w, void* packed, int Cout, int C0, int C1, int ksize, int wdtype,
                                    void* stream) {
  FD_REQUIRE(w && packed, "fd_conv_pack_weights: null pointer");
  FD_REQUIRE(ksize == 1 || ksize == 3, "fd_conv_pack_weights: ksize must be 1 or 3");
  FD_REQUIRE(wdtype == FD_BF16 || wdtype == FD_F32, "fd_conv_pack_weights: bad dtype");
  const int taps = ksize * ksize;
  const int CoutPad = cout_pad(Cout);
  const int CK = wdtype == FD_BF16 ? 32 : 16;
  const int nchunk0 = fd_cdiv(C0, CK), nchunks = nchunk0 + fd_cdiv(C1, CK);
  const long long total = (long long)nchunks * taps * CoutPad * CK;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  if (wdtype == FD_BF16)
    hipLaunchKernelGGL(pack_weights_kernel<bf16>, dim3(blocks), dim3(256), 0, fd_stream(stream), w,
                       reinterpret_cast<bf16*>(packed), Cout, CoutPad, C0, C1, taps, nchunk0, nchunks);
  else
    hipLaunchKernelGGL(pack_weights_kernel<float>, dim3(blocks), dim3(256), 0, fd_stream(stream), w,
                       reinterpret_cast<float*>(packed), Cout, CoutPad, C0, C1, taps, nchunk0, nchunks);
  FD_LAUNCH_CHECK();
  return FD_OK;
}

extern "C" int fd_conv2d(const void* in0, int C0, const void* in1, int C1, const float* affine, const void* packed_w,
                         const float* bias, int bias_rows, const void* skip, float scale, void* out, int Cout, int B,
                         int H, int W, int ksize, int dtype, int wdtype, void* stream) {
  FD_REQUIRE(in0 && packed_w && out, "fd_conv2d: null pointer");
  FD_REQUIRE(ksize == 1 || ksize == 3, "fd_conv2d: ksize must be 1 or 3 (got %d)", ksize);
  FD_REQUIRE(dtype == wdtype && (dtype == FD_BF16 || dtype == FD_F32),
             "fd_conv2d: supported modes are bf16 storage + bf16 MFMA, or f32 storage + f32 MFMA");
  FD_REQUIRE(C0 > 0 && C0 % 8 == 0 && C1 >= 0 && C1 % 8 == 0, "fd_conv2d: input channels must be multiples of 8 (C0=%d C1=%d)", C0, C1);
  FD_REQUIRE((C1 == 0) == (in1 == nullptr), "fd_conv2d: in1 / C1 mismatch");
  FD_REQUIRE(Cout > 0 && Cout % 4 == 0, "fd_conv2d: Cout must be a multiple of 4 (got %d)", Cout);
  FD_REQUIRE(B > 0 && H > 0 && W > 0, "fd_conv2d: bad shape");
  FD_REQUIRE(bias == nullptr || bias_rows == 1 || bias_rows == B, "fd_conv2d: bias_rows must be 1 or B");
  FD_REQUIRE((long long)B * H * W < (1ll << 31), "fd_conv2d: too many pixels for 32-bit indexing");
  FD_TRY(fd_conv_init_attributes());
  ConvArgs a{};
  a.in0 = in0; a.in1 = in1; a.C0 = C0; a.C1 = C1; a.affine = affine; a.w = packed_w; a.bias = bias;
  a.bias_rows = bias_rows; a.skip = skip; a.scale = scale; a.out = out; a.Cout = Cout; a.B = B; a.H = H; a.W = W;
  if (dtype == FD_BF16) return dispatch_conv<bf16>(a, ksize, fd_stream(stream));
  return dispatch_conv<float>(a, ksize, fd_stream(stream));
}
