// elementwise.hip -- the HBM-bound kernels of the NCSN++ vector field (NHWC activations):
// GroupNorm statistics, FIR x2 resampling (+ fused GroupNorm/SiLU), generic upfirdn2d,
// fused bias+act, time embedding, the 4-channel convs at the edges of the U-Net, and the
// ODE state updates fused with the final 1x1 output layer.
#include <type_traits>

#include "common.h"
#include "internal.h"

namespace {

// =====================================================================================================
// GroupNorm statistics: per-(b, c) sum and sum of squares (nn.GroupNorm, layerspp.py:229,241)
// =====================================================================================================
// grid (blocks_per_image, B), 256 threads = (C/8 channel vectors) x (2048/C pixel lanes).  Every block writes its
// own partial (sum, sum of squares) per channel: part[b][block][c][2] (float) -- deterministic, no atomics.  The
// MFMA conv epilogue emits the same format for its output (one partial per 16x16 tile).
template <typename T>
__global__ __launch_bounds__(256) void channel_sums_kernel(const T* __restrict__ x, float* __restrict__ part, int HW,
                                                           int C, int px_per_block) {
  const int cv = C >> 3;            // channel vectors (power of two, <= 32)
  const int lanes = 256 / cv;       // pixel lanes
  const int t = threadIdx.x;
  const int c8 = (t % cv) * 8, pl = t / cv;
  const int b = blockIdx.y;
  const int p0 = blockIdx.x * px_per_block;
  const int p1 = min(p0 + px_per_block, HW);
  float s[8], ss[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) s[i] = ss[i] = 0.f;
  const T* base = x + (size_t)b * HW * C + c8;
  for (int p = p0 + pl; p < p1; p += lanes) {
    float v[8];
    fd_load_vec<T, 8>(base + (size_t)p * C, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) { s[i] += v[i]; ss[i] = fmaf(v[i], v[i], ss[i]); }
  }
  __shared__ float red[256][17];
#pragma unroll
  for (int i = 0; i < 8; ++i) { red[t][i] = s[i]; red[t][8 + i] = ss[i]; }
  __syncthreads();
  for (int o = t; o < 2 * C; o += 256) {
    const int c = o >> 1, which = o & 1;
    const int tv = c >> 3, e = (c & 7) + 8 * which;
    float acc = 0.f;
    for (int l = 0; l < lanes; ++l) acc += red[l * cv + tv][e];
    part[(((size_t)b * gridDim.x + blockIdx.x) * C + c) * 2 + which] = acc;
  }
}

// One block per (group, b): reduce the partials of the group's channels over all tiles in double, then write the
// per-channel affine (a = rstd*gamma, d = beta - mean*rstd*gamma).  The group may straddle the two tensors of a
// virtual concat [C0 | C1] (e.g. GroupNorm(32, 320) over cat(h[256], hs[64]), ncsnpp.py:337).
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ p0, int tiles0, int stride0, int C0,
                                                          const float* __restrict__ p1, int tiles1, int stride1, int C1,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ affine, int groups, double inv_n, float eps) {
  const int C = C0 + C1, cpg = C / groups;
  const int g = blockIdx.x, b = blockIdx.y;
  const int c_lo = g * cpg;
  double s = 0.0, ss = 0.0;
  // channels of this group that live in tensor 0 / tensor 1
  const int a0 = c_lo < C0 ? c_lo : C0, a1 = (c_lo + cpg) < C0 ? (c_lo + cpg) : C0;          // [a0, a1) in tensor 0
  const int b0 = (c_lo > C0 ? c_lo : C0) - C0, b1 = ((c_lo + cpg) > C0 ? (c_lo + cpg) : C0) - C0;  // [b0, b1) in tensor 1
  const int n0 = a1 - a0, n1 = b1 - b0;
#pragma unroll 8   // independent loads in flight: the loop is latency-, not bandwidth-bound
  for (int i = threadIdx.x; i < n0 * tiles0; i += 256) {
    const int tl = i / n0, c = a0 + i % n0;
    const float2 v = *reinterpret_cast<const float2*>(p0 + (((size_t)b * tiles0 + tl) * stride0 + c) * 2);
    s += v.x; ss += v.y;
  }
#pragma unroll 8
  for (int i = threadIdx.x; i < n1 * tiles1; i += 256) {
    const int tl = i / n1, c = b0 + i % n1;
    const float2 v = *reinterpret_cast<const float2*>(p1 + (((size_t)b * tiles1 + tl) * stride1 + c) * 2);
    s += v.x; ss += v.y;
  }
  s = fd_wave_sum(s); ss = fd_wave_sum(ss);
  __shared__ double rs[4], rss[4];
  __shared__ float sh_mean_rstd[2];
  if ((threadIdx.x & 63) == 0) { rs[threadIdx.x >> 6] = s; rss[threadIdx.x >> 6] = ss; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double S = rs[0] + rs[1] + rs[2] + rs[3], SS = rss[0] + rss[1] + rss[2] + rss[3];
    const double mean = S * inv_n;
    double var = SS * inv_n - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    sh_mean_rstd[0] = (float)rstd;
    sh_mean_rstd[1] = (float)(mean * rstd);
  }
  __syncthreads();
  if (threadIdx.x < cpg) {
    const int c = c_lo + threadIdx.x;
    const float a = sh_mean_rstd[0] * gamma[c];
    affine[((size_t)b * C + c) * 2] = a;
    affine[((size_t)b * C + c) * 2 + 1] = beta[c] - sh_mean_rstd[1] * gamma[c];
  }
}

// silu(a*x + d), per-(b,c) affine: act(GroupNorm(x)) as a stand-alone pass (operator-level parity; fused on the hot path)
template <typename T>
__global__ __launch_bounds__(256) void gn_silu_apply_kernel(const T* __restrict__ x, const float* __restrict__ affine, T* __restrict__ out,
                                                            long long hw, int C, long long nvec) {
  const int cv = C >> 3;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
    const int c8 = (int)(i % cv) * 8;
    const long long b = i / cv / hw;
    float v[8];
    fd_load_vec<T, 8>(x + i * 8, v);
    const float* ad = affine + ((size_t)b * C + c8) * 2;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = fd_silu(fmaf(v[j], ad[2 * j], ad[2 * j + 1]));
    fd_store_vec<T, 8>(out + i * 8, v);
  }
}

// =====================================================================================================
// FIR [1,3,3,1] x2 resampling, polyphase form (up_or_down_sampling.py:220-282 via op/upfirdn2d.py)
//   down: out[n]   = (x[2n-1] + 3 x[2n] + 3 x[2n+1] + x[2n+2]) / 8     per axis, zeros outside
//   up  : out[2i]  = (x[i-1] + 3 x[i]) / 4 ; out[2i+1] = (3 x[i] + x[i+1]) / 4
// One thread = one VEC-channel vector of a strip of N consecutive rows at one column (output rows for down, input rows for
// up).  The filter is applied separably with a register sliding window DOWN the strip: every input row contributes one
// horizontally combined value per thread (3 / 4 neighbouring columns, activated once: silu(a*x+d), zero padding AFTER
// the activation), and the vertical taps run over the last 3 (up) / 4 (down) combined rows.  Per output this is 3x (up) /
// 2x (down) fewer loads and SiLU evaluations than the direct form, and at any moment neighbouring threads read
// neighbouring pixels of the same image row (contiguous HBM traffic).  With `affine`, the second output is the resample
// of the activated input.  N = 1 degenerates to the direct form (used without activation, where loads are all there is).
// =====================================================================================================
// The load is unconditional (the caller clamps the coordinates into the image, so the value is finite) and the zero
// padding is applied by a 0/1 factor: loads inside divergent branches would be waited for one at a time.
template <typename T, int VEC, bool ACT>
__device__ __forceinline__ void fir_load_px(const T* __restrict__ x, size_t base, bool ok, const float (&a)[VEC], const float (&d)[VEC],
                                            float (&r)[VEC], float (&ac)[VEC]) {
  fd_load_vec<T, VEC>(x + base, r);
  const float m = ok ? 1.f : 0.f;   // a multiply, not a select: hipcc turns the select into a divergent branch around the SiLU
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    ac[i] = ACT ? m * fd_silu(fmaf(r[i], a[i], d[i])) : 0.f;
    r[i] *= m;
  }
}
__device__ __forceinline__ int fir_clamp(int v, int n) { return v < 0 ? 0 : (v >= n ? n - 1 : v); }

// A pixel's VEC channels as loaded (not yet converted): the marching kernels request the NEXT row before they store the current one.
// Loads and stores return in order on this ISA (one vmcnt counter): a wait for loads issued AFTER a row's stores is a wait for the
// stores' acknowledgement as well -- measured on fir_up: 194 us of loads + arithmetic and 237 us of stores ran back to back (431 us).
template <typename T, int VEC> struct fir_raw;
template <> struct fir_raw<bf16, 8> { bf16x8 v; __device__ void load(const bf16* p) { v = *reinterpret_cast<const bf16x8*>(p); }
  __device__ void get(float (&f)[8]) const {
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)v[i]; } };
template <> struct fir_raw<bf16, 4> { bf16x4 v; __device__ void load(const bf16* p) { v = *reinterpret_cast<const bf16x4*>(p); }
  __device__ void get(float (&f)[4]) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = (float)v[i]; } };
template <> struct fir_raw<float, 4> { f32x4 v; __device__ void load(const float* p) { v = *reinterpret_cast<const f32x4*>(p); }
  __device__ void get(float (&f)[4]) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = v[i]; } };
// activation + zero padding of an already loaded pixel (the arithmetic of fir_load_px)
template <int VEC, bool ACT>
__device__ __forceinline__ void fir_act_px(bool ok, const float (&a)[VEC], const float (&d)[VEC], float (&r)[VEC], float (&ac)[VEC]) {
  const float m = ok ? 1.f : 0.f;
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    ac[i] = ACT ? m * fd_silu(fmaf(r[i], a[i], d[i])) : 0.f;
    r[i] *= m;
  }
}

// down: one thread = one VEC-channel vector of a BY x BX block of output pixels: the (2BY+2) x (2BX+2) input patch is
// loaded and activated once (2x2: 9 instead of 16 SiLU evaluations per output; 1x1 is the direct form).
template <typename T, int VEC, bool ACT, int BY, int BX>
__global__ __launch_bounds__(256) void fir_down_kernel(const T* __restrict__ x, const float* __restrict__ affine,
                                                       T* __restrict__ out_raw, T* __restrict__ out_act, int B, int H,
                                                       int W, int C) {
  constexpr int PSY = 2 * BY + 2, PSX = 2 * BX + 2;   // input patch
  const int cvn = C / VEC;
  const int OH = H >> 1, OW = W >> 1, nbx = (OW + BX - 1) / BX, nby = (OH + BY - 1) / BY;
  const long long total = (long long)B * nby * nbx * cvn;
  const long long idx = blockIdx.x * 256ll + threadIdx.x;
  if (idx >= total) return;
  const int cv = (int)(idx % cvn);
  long long q = idx / cvn;
  const int ox0 = (int)(q % nbx) * BX; q /= nbx;
  const int oy0 = (int)(q % nby) * BY;
  const int b = (int)(q / nby);
  const int c = cv * VEC;
  float a[VEC], d[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { a[i] = 0.f; d[i] = 0.f; }
  if (ACT) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) { a[i] = affine[((size_t)b * C + c + i) * 2]; d[i] = affine[((size_t)b * C + c + i) * 2 + 1]; }
  }
  const float wt[4] = {0.125f, 0.375f, 0.375f, 0.125f};
  float accr[BY][BX][VEC], acca[BY][BX][VEC];
#pragma unroll
  for (int py = 0; py < BY; ++py)
#pragma unroll
    for (int px = 0; px < BX; ++px)
#pragma unroll
      for (int i = 0; i < VEC; ++i) { accr[py][px][i] = 0.f; acca[py][px][i] = 0.f; }
  const size_t img = (size_t)b * H * W;
#pragma unroll
  for (int ry = 0; ry < PSY; ++ry) {               // input row iy = 2 * oy0 - 1 + ry
    const int iy = 2 * oy0 - 1 + ry;
    const bool yok = iy >= 0 && iy < H;
    float hr[BX][VEC], ha[BX][VEC];               // horizontal sums of this row for the BX output columns
#pragma unroll
    for (int px = 0; px < BX; ++px)
#pragma unroll
      for (int i = 0; i < VEC; ++i) { hr[px][i] = 0.f; ha[px][i] = 0.f; }
#pragma unroll
    for (int rx = 0; rx < PSX; ++rx) {
      const int ix = 2 * ox0 - 1 + rx;
      float r[VEC], ac[VEC];
      fir_load_px<T, VEC, ACT>(x, (img + (size_t)fir_clamp(iy, H) * W + fir_clamp(ix, W)) * C + c, yok && ix >= 0 && ix < W, a, d, r, ac);
#pragma unroll
      for (int px = 0; px < BX; ++px) {
        const int kx = rx - 2 * px;               // tap index of this column for output column px
        if (kx >= 0 && kx < 4) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) { hr[px][i] = fmaf(wt[kx], r[i], hr[px][i]); if (ACT) ha[px][i] = fmaf(wt[kx], ac[i], ha[px][i]); }
        }
      }
    }
#pragma unroll
    for (int py = 0; py < BY; ++py) {
      const int ky = ry - 2 * py;
      if (ky >= 0 && ky < 4) {
#pragma unroll
        for (int px = 0; px < BX; ++px)
#pragma unroll
          for (int i = 0; i < VEC; ++i) {
            accr[py][px][i] = fmaf(wt[ky], hr[px][i], accr[py][px][i]);
            if (ACT) acca[py][px][i] = fmaf(wt[ky], ha[px][i], acca[py][px][i]);
          }
      }
    }
    __builtin_amdgcn_sched_barrier(0);            // one patch row of loads in flight at a time (bounds the register use)
  }
#pragma unroll
  for (int py = 0; py < BY; ++py)
#pragma unroll
    for (int px = 0; px < BX; ++px) {
      const int oy = oy0 + py, ox = ox0 + px;
      if (oy < OH && ox < OW) {
        const size_t o = (((size_t)b * OH + oy) * OW + ox) * C + c;
        if (out_raw) fd_store_vec<T, VEC>(out_raw + o, accr[py][px]);
        if (ACT && out_act) fd_store_vec<T, VEC>(out_act + o, acca[py][px]);
      }
    }
}

// down, big grids: one thread = 4 channels of a BX-wide column block, MARCHING down a strip of NR output rows: every input row is
// loaded and activated once per thread and enters the two output rows it belongs to ((2BX+2)/(2BX) x (2NR+2)/(2NR) = 1.33 SiLU
// evaluations per input instead of 1.875 with 4x2 blocks).  Per output the fma sequence is the one of fir_down_kernel (horizontal taps
// ascending from 0, then vertical taps ascending from 0): identical bits.  The FIR arithmetic is written on channel pairs (v_pk_fma_f32).
typedef float fir_f2 __attribute__((ext_vector_type(2)));
// FAST (every strip and column block full, both outputs): the stores are unconditional and the row-pair step exists three times --
// without stores (row pair 0), with stores (row pair 1, peeled) and as the loop body -- so that every path into a wait for the next
// input row carries the same memory operations: hipcc's s_waitcnt vmcnt(n) then counts the stores still in flight instead of waiting
// for them (see fir_raw).
template <typename T, int BX, int NR, bool FAST = false>
__global__ __launch_bounds__(256) void fir_down_march_kernel(const T* __restrict__ x, const float* __restrict__ affine, T* __restrict__ out_raw,
                                                             T* __restrict__ out_act, int B, int H, int W, int C) {
  constexpr int VEC = 4, PSX = 2 * BX + 2;
  const int cvn = C / VEC;
  const int OH = H >> 1, OW = W >> 1, nbx = (OW + BX - 1) / BX, nst = (OH + NR - 1) / NR;
  const long long total = (long long)B * nst * nbx * cvn;
  const long long idx = blockIdx.x * 256ll + threadIdx.x;
  if (idx >= total) return;
  const int cv = (int)(idx % cvn);
  long long q = idx / cvn;
  const int ox0 = (int)(q % nbx) * BX; q /= nbx;
  const int oy0 = (int)(q % nst) * NR;
  const int b = (int)(q / nst);
  const int c = cv * VEC;
  float a[VEC], d[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { a[i] = affine[((size_t)b * C + c + i) * 2]; d[i] = affine[((size_t)b * C + c + i) * 2 + 1]; }
  const fir_f2 wt2[4] = {{0.125f, 0.125f}, {0.375f, 0.375f}, {0.375f, 0.375f}, {0.125f, 0.125f}};
  // [output row: prev = rp - 1 | cur = rp][column][raw lo, raw hi, act lo, act hi channel pairs]
  fir_f2 acc[2][BX][4];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int px = 0; px < BX; ++px)
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[s][px][k] = fir_f2{0.f, 0.f};
  const size_t img = (size_t)b * H * W;
  int xi[PSX]; bool xok[PSX];
#pragma unroll
  for (int rx = 0; rx < PSX; ++rx) { const int ix = 2 * ox0 - 1 + rx; xok[rx] = ix >= 0 && ix < W; xi[rx] = fir_clamp(ix, W); }
  const int nrows = FAST ? NR : (OH - oy0 < NR ? OH - oy0 : NR);
  fir_raw<T, VEC> nxt[PSX];
  auto request_row = [&](int iy) {              // (clamped: the zero padding is a factor later)
    const size_t row = (img + (size_t)fir_clamp(iy, H) * W) * C + c;
#pragma unroll
    for (int rx = 0; rx < PSX; ++rx) nxt[rx].load(x + row + (size_t)xi[rx] * C);
  };
  // row pair rp = input rows 2 (oy0 + rp) - 1 and 2 (oy0 + rp): taps (2, 3) of output row rp - 1, (0, 1) of rp; then output row rp - 1 is
  // complete (stored when `store`), and rp becomes the previous row
  auto step = [&](int rp, auto store) __attribute__((always_inline)) {
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      const int iy = 2 * (oy0 + rp) - 1 + par;
      const bool yok = iy >= 0 && iy < H;
      float r[PSX][VEC];
#pragma unroll
      for (int rx = 0; rx < PSX; ++rx) nxt[rx].get(r[rx]);
      request_row(iy + 1);                      // the next row, before this row pair's stores; one row past the strip at the end
      __builtin_amdgcn_sched_barrier(0);
      fir_f2 h[BX][4];
#pragma unroll
      for (int px = 0; px < BX; ++px)
#pragma unroll
        for (int k = 0; k < 4; ++k) h[px][k] = fir_f2{0.f, 0.f};
#pragma unroll
      for (int rx = 0; rx < PSX; ++rx) {
        const float m = yok && xok[rx] ? 1.f : 0.f;
        float ac[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) { ac[i] = m * fd_silu(fmaf(r[rx][i], a[i], d[i])); r[rx][i] *= m; }
        const fir_f2 v[4] = {{r[rx][0], r[rx][1]}, {r[rx][2], r[rx][3]}, {ac[0], ac[1]}, {ac[2], ac[3]}};
#pragma unroll
        for (int px = 0; px < BX; ++px) {
          const int kx = rx - 2 * px;
          if (kx >= 0 && kx < 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) h[px][k] = __builtin_elementwise_fma(wt2[kx], v[k], h[px][k]);
          }
        }
      }
#pragma unroll
      for (int px = 0; px < BX; ++px)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          acc[1][px][k] = __builtin_elementwise_fma(wt2[par], h[px][k], acc[1][px][k]);
          acc[0][px][k] = __builtin_elementwise_fma(wt2[2 + par], h[px][k], acc[0][px][k]);
        }
    }
    if (decltype(store)::value) {
      const int oy = oy0 + rp - 1;
#pragma unroll
      for (int px = 0; px < BX; ++px) {
        const int ox = ox0 + px;
        if (FAST || ox < OW) {
          const size_t o = (((size_t)b * OH + oy) * OW + ox) * C + c;
          const float vr[VEC] = {acc[0][px][0][0], acc[0][px][0][1], acc[0][px][1][0], acc[0][px][1][1]};
          const float va[VEC] = {acc[0][px][2][0], acc[0][px][2][1], acc[0][px][3][0], acc[0][px][3][1]};
          if (FAST || out_raw) fd_store_vec<T, VEC>(out_raw + o, vr);
          if (FAST || out_act) fd_store_vec<T, VEC>(out_act + o, va);
        }
      }
    }
#pragma unroll
    for (int px = 0; px < BX; ++px)
#pragma unroll
      for (int k = 0; k < 4; ++k) { acc[0][px][k] = acc[1][px][k]; acc[1][px][k] = fir_f2{0.f, 0.f}; }
  };
  request_row(2 * oy0 - 1);
  step(0, std::false_type{});
  if (FAST) {
    step(1, std::true_type{});
#pragma unroll 1
    for (int rp = 2; rp <= NR; ++rp) step(rp, std::true_type{});
  } else {
#pragma unroll 1
    for (int rp = 1; rp <= nrows; ++rp) step(rp, std::true_type{});
  }
}

// up: one thread = VEC channels of BX input columns, N input rows: 2 BX output columns x 2N output rows; BX + 2 columns are loaded and
// activated per row ((BX + 2) / BX x (N + 2) / N activations per input: 3.75 at 1 x 8, 2.5 at 2 x 8).  The arithmetic is written on
// channel pairs (v_pk_mul_f32 / v_pk_fma_f32: the same IEEE operations per element as the scalar form); every output is the same
// operation sequence for every (N, BX).
// FAST: every strip is full (H % N == 0, W % BX == 0) and both outputs exist: the stores are UNCONDITIONAL -- no control flow between a
// row's loads and the next row's use, so hipcc's s_waitcnt vmcnt(n) counts the stores in flight exactly instead of joining a path
// without them (which made every wait for a row also a wait for the previous row's stores).
template <typename T, int VEC, bool ACT, int N, int BX, bool FAST = false>
__global__ __launch_bounds__(256) void fir_up_kernel(const T* __restrict__ x, const float* __restrict__ affine,
                                                     T* __restrict__ out_raw, T* __restrict__ out_act, int B, int H,
                                                     int W, int C) {
  constexpr int NP = VEC / 2;
  const int cvn = C / VEC, ns = (H + N - 1) / N, nbx = (W + BX - 1) / BX;
  const long long total = (long long)B * ns * nbx * cvn;
  const long long idx = blockIdx.x * 256ll + threadIdx.x;
  if (idx >= total) return;
  const int cv = (int)(idx % cvn);
  long long q = idx / cvn;
  const int ix0 = (int)(q % nbx) * BX; q /= nbx;
  const int y0 = (int)(q % ns) * N;
  const int b = (int)(q / ns);
  const int c = cv * VEC;
  float a[VEC], d[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { a[i] = 0.f; d[i] = 0.f; }
  if (ACT) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) { a[i] = affine[((size_t)b * C + c + i) * 2]; d[i] = affine[((size_t)b * C + c + i) * 2 + 1]; }
  }
  const int OW = 2 * W, OH = 2 * H;
  const fir_f2 w75 = {0.75f, 0.75f}, w25 = {0.25f, 0.25f};
  // horizontally combined rows for the output columns 2ix (px = 0: (x[ix-1] + 3 x[ix]) / 4) and 2ix+1 (px = 1), slot = row % 3
  fir_f2 hr[3][2 * BX][NP], ha[3][2 * BX][NP];
  const size_t img = (size_t)b * H * W;
  fir_raw<T, VEC> nxt[BX + 2];
  auto request_row = [&](int j) {                // input row y0 - 1 + j (clamped: the zero padding is a factor later)
    const size_t row = (img + (size_t)fir_clamp(y0 - 1 + j, H) * W) * C + c;
#pragma unroll
    for (int dx = 0; dx < BX + 2; ++dx) nxt[dx].load(x + row + (size_t)fir_clamp(ix0 + dx - 1, W) * C);
  };
  request_row(0);
#pragma unroll
  for (int j = 0; j < N + 2; ++j) {              // input row yy = y0 - 1 + j
    const int yy = y0 - 1 + j;
    const bool yok = yy >= 0 && yy < H;
    float r[BX + 2][VEC], ac[BX + 2][VEC];
#pragma unroll
    for (int dx = 0; dx < BX + 2; ++dx) nxt[dx].get(r[dx]);
    if (j + 1 < N + 2) request_row(j + 1);       // before this row's stores (see fir_raw)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int dx = 0; dx < BX + 2; ++dx) {
      const int xx = ix0 + dx - 1;
      fir_act_px<VEC, ACT>(yok && xx >= 0 && xx < W, a, d, r[dx], ac[dx]);
    }
#pragma unroll
    for (int bx = 0; bx < BX; ++bx)
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const fir_f2 r0 = {r[bx][2 * i], r[bx][2 * i + 1]}, r1 = {r[bx + 1][2 * i], r[bx + 1][2 * i + 1]}, r2 = {r[bx + 2][2 * i], r[bx + 2][2 * i + 1]};
        hr[j % 3][2 * bx][i] = __builtin_elementwise_fma(w75, r1, w25 * r0);
        hr[j % 3][2 * bx + 1][i] = __builtin_elementwise_fma(w75, r1, w25 * r2);
        if (ACT) {
          const fir_f2 a0 = {ac[bx][2 * i], ac[bx][2 * i + 1]}, a1 = {ac[bx + 1][2 * i], ac[bx + 1][2 * i + 1]}, a2 = {ac[bx + 2][2 * i], ac[bx + 2][2 * i + 1]};
          ha[j % 3][2 * bx][i] = __builtin_elementwise_fma(w75, a1, w25 * a0);
          ha[j % 3][2 * bx + 1][i] = __builtin_elementwise_fma(w75, a1, w25 * a2);
        }
      }
    if (j >= 2) {                                // rows j-2, j-1, j complete the two output rows of input row iy = yy - 1
      const int iy = yy - 1;
      if (FAST || iy < H) {
        const int up = (j - 2) % 3, m = (j - 1) % 3, dn = j % 3;
#pragma unroll
        for (int py = 0; py < 2; ++py) {
          const int other = py == 0 ? up : dn;   // py = 0: (row[iy-1] + 3 row[iy]) / 4; py = 1: (3 row[iy] + row[iy+1]) / 4
          const size_t o = (((size_t)b * OH + 2 * iy + py) * OW + 2 * ix0) * C + c;
#pragma unroll
          for (int px = 0; px < 2 * BX; ++px) {
            if (!FAST && BX > 1 && ix0 + px / 2 >= W) continue;
            float o0[VEC], p0[VEC];
#pragma unroll
            for (int i = 0; i < NP; ++i) {
              const fir_f2 t0 = __builtin_elementwise_fma(w75, hr[m][px][i], w25 * hr[other][px][i]);
              o0[2 * i] = t0[0]; o0[2 * i + 1] = t0[1];
              if (ACT) {
                const fir_f2 u0 = __builtin_elementwise_fma(w75, ha[m][px][i], w25 * ha[other][px][i]);
                p0[2 * i] = u0[0]; p0[2 * i + 1] = u0[1];
              }
            }
            if (FAST || out_raw) fd_store_vec<T, VEC>(out_raw + o + (size_t)px * C, o0);
            if (ACT && (FAST || out_act)) fd_store_vec<T, VEC>(out_act + o + (size_t)px * C, p0);
          }
        }
      }
    }
  }
}

// =====================================================================================================
// Generic upfirdn2d (op/upfirdn2d.py:183-224 semantics; CUDA twin upfirdn2d_kernel.cu:60-218)
// =====================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void upfirdn2d_kernel(const T* __restrict__ in, const float* __restrict__ kernel,
                                                        T* __restrict__ out, int major, int in_h, int in_w, int minor,
                                                        int kh, int kw, int up_x, int up_y, int down_x, int down_y,
                                                        int pad_x0, int pad_y0, int out_h, int out_w) {
  const long long total = (long long)major * out_h * out_w * minor;
  for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int mi = (int)(idx % minor);
    long long r = idx / minor;
    const int ox = (int)(r % out_w); r /= out_w;
    const int oy = (int)(r % out_h);
    const int mj = (int)(r / out_h);
    float acc = 0.f;
    // out[oy,ox] = sum_{i,j} kernel[kh-1-i][kw-1-j] * U[oy*down_y + i - pad_y0][ox*down_x + j - pad_x0],
    // U = zero-inserted input (U[u][v] = in[u/up_y][v/up_x] when both divisible and in range)
    for (int i = 0; i < kh; ++i) {
      const int u = oy * down_y + i - pad_y0;
      if (u < 0 || u % up_y != 0) continue;
      const int iy = u / up_y;
      if (iy >= in_h) continue;
      for (int j = 0; j < kw; ++j) {
        const int v = ox * down_x + j - pad_x0;
        if (v < 0 || v % up_x != 0) continue;
        const int ix = v / up_x;
        if (ix >= in_w) continue;
        acc = fmaf(kernel[(kh - 1 - i) * kw + (kw - 1 - j)], Elem<T>::ld(in + (((size_t)mj * in_h + iy) * in_w + ix) * minor + mi), acc);
      }
    }
    Elem<T>::st(out + idx, acc);
  }
}

// fused_bias_act_kernel.cu:30-61, grad == 0
__global__ void fused_bias_act_kernel(const float* __restrict__ x, const float* __restrict__ bias, float* __restrict__ out,
                                      long long n, int step_b, int size_b, int act, float alpha, float scale) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float v = x[i];
    if (bias) v += bias[(i / step_b) % size_b];
    if (act == 3) v = v > 0.f ? v : v * alpha;
    out[i] = v * scale;
  }
}

// =====================================================================================================
// Time embedding (ncsnpp.py:263-274, layerspp.py:42-51) -- one block per t row.
// =====================================================================================================
// acc = init + sum_k w[k] * x[k], k ascending (one fma per term: the order the parity tests pin), n % 4 == 0.  The weight row is read
// as 16-byte vectors with 16 of them in flight: these matrix-vector products are one dependent chain per thread and sit on the
// critical path of every network evaluation.
__device__ __forceinline__ float fd_row_dot(const float* __restrict__ w, const float* x, int n, float init) {
  float acc = init;
  for (int k0 = 0; k0 < n; k0 += 64) {
    f32x4 wv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) wv[j] = (k0 + 4 * j < n) ? *reinterpret_cast<const f32x4*>(w + k0 + 4 * j) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (k0 + 4 * j < n) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = fmaf(wv[j][e], x[k0 + 4 * j + e], acc);
      }
  }
  return acc;
}

__global__ __launch_bounds__(256) void time_embedding_kernel(const float* __restrict__ t, float t_imm, const float* __restrict__ W,
                                                             int nf, const float* __restrict__ w1, const float* __restrict__ b1,
                                                             const float* __restrict__ w2, const float* __restrict__ b2,
                                                             float* __restrict__ temb) {
  extern __shared__ float sm[];
  float* emb = sm;            // [2nf]
  float* hid = sm + 2 * nf;   // [4nf]
  const int r = blockIdx.x;
  const float tv = t ? t[r] : t_imm;
  for (int j = threadIdx.x; j < nf; j += blockDim.x) {
    const float xp = ((tv * W[j]) * 2.0f) * 3.14159274101257324f;  // x[:,None]*W[None,:]*2*np.pi in float32
    emb[j] = sinf(xp);
    emb[nf + j] = cosf(xp);
  }
  __syncthreads();
  const int E = 2 * nf, D = 4 * nf;
  for (int o = threadIdx.x; o < D; o += blockDim.x) {
    hid[o] = fd_silu(fd_row_dot(w1 + (size_t)o * E, emb, E, b1[o]));
  }
  __syncthreads();
  for (int o = threadIdx.x; o < D; o += blockDim.x) {
    temb[(size_t)r * D + o] = fd_row_dot(w2 + (size_t)o * D, hid, D, b2[o]);
  }
}

// out[r][o] = cb[o] + db[o] + sum_k dw[o][k] * silu(temb[r][k]); one job per ResnetBlock (layerspp.py:272-273)
__global__ __launch_bounds__(256) void temb_bias_kernel(const fd_temb_job* __restrict__ jobs, const float* __restrict__ temb,
                                                        int temb_dim) {
  extern __shared__ float st[];  // silu(temb[r])
  const fd_temb_job job = jobs[blockIdx.x];
  const int r = blockIdx.y;
  for (int k = threadIdx.x; k < temb_dim; k += blockDim.x) st[k] = fd_silu(temb[(size_t)r * temb_dim + k]);
  __syncthreads();
  for (int o = threadIdx.x; o < job.Cout; o += blockDim.x) {
    job.out[(size_t)r * job.Cout + o] = fd_row_dot(job.dense_w + (size_t)o * temb_dim, st, temb_dim, job.dense_b[o]) + job.conv_b[o];
  }
}

__global__ __launch_bounds__(256) void temb_bias_single_kernel(fd_temb_job job, const float* __restrict__ temb, int temb_dim) {
  extern __shared__ float st[];
  const int r = blockIdx.x;
  for (int k = threadIdx.x; k < temb_dim; k += blockDim.x) st[k] = fd_silu(temb[(size_t)r * temb_dim + k]);
  __syncthreads();
  for (int o = threadIdx.x; o < job.Cout; o += blockDim.x) {
    job.out[(size_t)r * job.Cout + o] = fd_row_dot(job.dense_w + (size_t)o * temb_dim, st, temb_dim, job.dense_b[o]) + (job.conv_b ? job.conv_b[o] : 0.f);
  }
}

// =====================================================================================================
// Edge-of-network kernels on the 4-channel tensors
// =====================================================================================================
// cat(x.re, x.im, y.re, y.im) (ncsnpp.py:401-404) -> NHWC [B][F][T][8]; channels 4..7 are zero padding so that the
// tensor (and its FIR-downsampled pyramid) can feed the MFMA conv, whose input channel counts are multiples of 8
template <typename T>
__global__ void pack_input_kernel(const float2* __restrict__ x, const float2* __restrict__ y, T* __restrict__ out, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float2 a = x[i], b = y[i];
    const float v[8] = {a.x, a.y, b.x, b.y, 0.f, 0.f, 0.f, 0.f};
    fd_store_vec<T, 8>(out + 8 * i, v);
  }
}

// The input convolution all_modules.3 (conv3x3, 4 -> nf; ncsnpp.py:291, layers.py:128-134) with the GroupNorm partial sums of its output.
// On the matrix cores this layer is all prologue and epilogue (K = 4 padded channels x 9 taps: 237 us at 8 x 768 x 256 for a launch that
// writes 201 MB); here it is f32 vector FMAs next to the stores: one block = a 16 x 16 pixel tile, one thread = 8 couts x NG = Cout / 8
// consecutive pixels of a tile row, taps ascending, input channels ascending, one fma each (deterministic; statistics of the f32 values,
// fixed reduction order).  Input = the packed NHWC tensor of pack_input_kernel (4 real channels of 8), zero padding.
typedef float f32x2e __attribute__((ext_vector_type(2)));
typedef float f32x4e __attribute__((ext_vector_type(4)));
template <typename T, int NG>
__global__ __launch_bounds__(256) void conv_in_kernel(const T* __restrict__ in8, const float* __restrict__ w, const float* __restrict__ bias,
                                                      T* __restrict__ out, float* __restrict__ part, int H, int W) {
  constexpr int C = 8 * NG;
  constexpr int WBLK = 9;                         // f32x4 per (tap, cout group): 4 channels x 2 cout halves + 1 pad (bank spread)
  __shared__ f32x4e halo[18][18];
  __shared__ f32x4e wl[9 * NG * WBLK];
  __shared__ float red[4][C][2];
  const int t = threadIdx.x, tile = blockIdx.x, b = blockIdx.y;
  const int tiles_w = W >> 4;
  const int h0 = (tile / tiles_w) << 4, w0 = (tile % tiles_w) << 4;
  for (int i = t; i < 18 * 18; i += 256) {
    const int hr = i / 18, hc = i - hr * 18;
    const int gh = h0 - 1 + hr, gw = w0 - 1 + hc;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (gh >= 0 && gh < H && gw >= 0 && gw < W) fd_load_vec<T, 4>(in8 + (((size_t)b * H + gh) * W + gw) * 8, v);
    halo[hr][hc] = f32x4e{v[0], v[1], v[2], v[3]};
  }
  for (int e = t; e < C * 36; e += 256) {         // w: [Cout][4][3][3]
    const int co = e / 36, r = e - co * 36, ci = r / 9, tap = r - ci * 9;
    reinterpret_cast<float*>(wl)[(((tap * NG + (co >> 3)) * WBLK + ci * 2 + ((co & 7) >> 2)) << 2) + (co & 3)] = w[e];
  }
  __syncthreads();
  const int cg = t % NG, slot = t / NG;
  const int p0 = slot * NG, r = p0 >> 4, c0 = p0 & 15;
  f32x2e acc[NG][4];
#pragma unroll
  for (int p = 0; p < NG; ++p)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[p][j] = f32x2e{bias[cg * 8 + 2 * j], bias[cg * 8 + 2 * j + 1]};
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int dy = tap / 3, dx = tap % 3;
    f32x4e wv[4][2];
#pragma unroll
    for (int ci = 0; ci < 4; ++ci)
#pragma unroll
      for (int h = 0; h < 2; ++h) wv[ci][h] = wl[(tap * NG + cg) * WBLK + ci * 2 + h];
#pragma unroll
    for (int p = 0; p < NG; ++p) {
      const f32x4e x = halo[r + dy][c0 + p + dx];
#pragma unroll
      for (int ci = 0; ci < 4; ++ci) {
        const f32x2e xx = {x[ci], x[ci]};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x2e w2 = {wv[ci][j >> 1][2 * (j & 1)], wv[ci][j >> 1][2 * (j & 1) + 1]};
          acc[p][j] = __builtin_elementwise_fma(xx, w2, acc[p][j]);
        }
      }
    }
  }
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
#pragma unroll
  for (int p = 0; p < NG; ++p) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[2 * j] = acc[p][j][0]; v[2 * j + 1] = acc[p][j][1]; }
    fd_store_vec<T, 8>(out + (((size_t)b * H + h0 + r) * W + w0 + c0 + p) * C + cg * 8, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] += v[j]; s2[j] = fmaf(v[j], v[j], s2[j]); }
  }
  // lanes of a wave that share a cout group differ in the slot bits: fold them, then the four waves through LDS
#pragma unroll
  for (int msk = NG; msk < 64; msk <<= 1)
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] += __shfl_xor(s1[j], msk, 64); s2[j] += __shfl_xor(s2[j], msk, 64); }
  if ((t & 63) < NG) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[t >> 6][cg * 8 + j][0] = s1[j]; red[t >> 6][cg * 8 + j][1] = s2[j]; }
  }
  __syncthreads();
  if (t < C * 2) {
    const int c = t >> 1, which = t & 1;
    part[(((size_t)b * gridDim.x + tile) * C + c) * 2 + which] = (red[0][c][which] + red[1][c][which]) + (red[2][c][which] + red[3][c][which]);
  }
}

// Combine 'sum' (layerspp.py:54-69): out = conv1x1(p4) + bias + h, fused with the per-tile GroupNorm partial sums of `out`
// (same [b][tile][C][2] format as channel_sums_kernel; the next ResnetBlock's GroupNorm_0 consumes them).  One block =
// COMBINE_PX pixels of one image; one thread = 8 couts of every (256 / (Cout / 8))-th pixel.
constexpr int COMBINE_PX = 256;
template <typename T>
__global__ __launch_bounds__(256) void combine_kernel(const T* __restrict__ p4, const float* __restrict__ w, const float* __restrict__ bias,
                                                      const T* __restrict__ h, T* __restrict__ out, float* __restrict__ part, int HW, int Cout) {
  const int cv = Cout >> 3, lanes = 256 / cv;
  const int t = threadIdx.x, cg = t % cv, pl = t / cv;
  const int b = blockIdx.y;
  const int p0 = blockIdx.x * COMBINE_PX, p1 = min(p0 + COMBINE_PX, HW);
  float wr[8][4], bs[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    bs[i] = bias[cg * 8 + i];
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) wr[i][ci] = w[(size_t)(cg * 8 + i) * 4 + ci];
  }
  float s[8], ss[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) s[i] = ss[i] = 0.f;
  // eight pixels per trip with all loads issued first (the statistics are accumulated in the same pixel order as a plain loop)
  constexpr int UP = 8;
  for (int pb = p0 + pl; pb < p1; pb += UP * lanes) {
    float v[UP][4], hv[UP][8];
#pragma unroll
    for (int u = 0; u < UP; ++u) {
      const int p = pb + u * lanes;
      const size_t pix = (size_t)b * HW + (p < p1 ? p : p0);
      fd_load_vec<T, 4>(p4 + pix * 8, v[u]);   // the 4-channel pyramid is stored with 8-channel stride (see pack_input_kernel)
      fd_load_vec<T, 8>(h + pix * Cout + cg * 8, hv[u]);
    }
#pragma unroll
    for (int u = 0; u < UP; ++u) {
      const int p = pb + u * lanes;
      if (p < p1) {
        const size_t pix = (size_t)b * HW + p;
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float acc = bs[i];
#pragma unroll
          for (int ci = 0; ci < 4; ++ci) acc = fmaf(wr[i][ci], v[u][ci], acc);
          o[i] = acc + hv[u][i];
          const float q = (float)(T)o[i];     // statistics of the stored (rounded) tensor, as channel_sums_kernel would see it
          s[i] += q; ss[i] = fmaf(q, q, ss[i]);
        }
        fd_store_vec<T, 8>(out + pix * Cout + cg * 8, o);
      }
    }
  }
  __shared__ float red[256][17];
#pragma unroll
  for (int i = 0; i < 8; ++i) { red[t][i] = s[i]; red[t][8 + i] = ss[i]; }
  __syncthreads();
  for (int o = t; o < 2 * Cout; o += 256) {
    const int c = o >> 1, which = o & 1;
    const int tv = c >> 3, e = (c & 7) + 8 * which;
    float acc = 0.f;
    for (int l = 0; l < lanes; ++l) acc += red[l * cv + tv][e];
    part[(((size_t)b * gridDim.x + blockIdx.x) * Cout + c) * 2 + which] = acc;
  }
}

// output_layer (1x1, 4 -> 2, no bias; ncsnpp.py:100,398) + view_as_complex (:407-411) fused with the
// solver's state update:  dst = base + coef * (v + k_old);  optionally k_save = v.
template <typename T>
__global__ void output_update_kernel(const T* __restrict__ pyr, const float* __restrict__ wo, const float2* __restrict__ base,
                                     const float2* __restrict__ kold, float coef, float2* __restrict__ dst,
                                     float2* __restrict__ ksave, long long n) {
  const float w0 = wo[0], w1 = wo[1], w2 = wo[2], w3 = wo[3], w4 = wo[4], w5 = wo[5], w6 = wo[6], w7 = wo[7];
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float p[4];
    fd_load_vec<T, 4>(pyr + 4 * i, p);
    float2 v;
    v.x = fmaf(w3, p[3], fmaf(w2, p[2], fmaf(w1, p[1], w0 * p[0])));
    v.y = fmaf(w7, p[3], fmaf(w6, p[2], fmaf(w5, p[1], w4 * p[0])));
    if (ksave) ksave[i] = v;
    float2 s = v;
    if (kold) { const float2 k = kold[i]; s.x += k.x; s.y += k.y; }
    float2 o = {coef * s.x, coef * s.y};
    if (base) { const float2 bb = base[i]; o.x += bb.x; o.y += bb.y; }
    dst[i] = o;
  }
}

// x0 = Y + sigma_fac * (sigma_y[f] * noise).type(complex64)   (model.py:512, :530-536); sigma is float64
__global__ void init_state_kernel(const float2* __restrict__ Y, const float2* __restrict__ noise, const double* __restrict__ sigma,
                                  int sigma_n, float sigma_fac, float2* __restrict__ x0, int F, int T, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int f = (int)((i / T) % F);
    const double s = sigma[sigma_n == 1 ? 0 : f];
    const float2 nz = noise[i], yv = Y[i];
    const float nx = (float)(s * (double)nz.x), ny = (float)(s * (double)nz.y);
    float2 o = {yv.x + sigma_fac * nx, yv.y + sigma_fac * ny};
    x0[i] = o;
  }
}

// Predictor / corrector update of the ScoreDec sampler (sampling/predictors.py:48-71, correctors.py:52-66) fused with the
// output layer:  dst = cb * base + cy * Y + cn * v + cz * z   (v = output_layer(pyr) as complex; every coefficient is a
// real scalar that the host derives from the OUVE closed forms, sdes.py:168-192).
template <typename T>
__global__ void score_update_kernel(const T* __restrict__ pyr, const float* __restrict__ wo, const float2* __restrict__ base, float cb,
                                    const float2* __restrict__ Y, float cy, float cn, const float2* __restrict__ z, float cz,
                                    float2* __restrict__ dst, long long n) {
  const float w0 = wo[0], w1 = wo[1], w2 = wo[2], w3 = wo[3], w4 = wo[4], w5 = wo[5], w6 = wo[6], w7 = wo[7];
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float p[4];
    fd_load_vec<T, 4>(pyr + 4 * i, p);
    const float vx = fmaf(w3, p[3], fmaf(w2, p[2], fmaf(w1, p[1], w0 * p[0])));
    const float vy = fmaf(w7, p[3], fmaf(w6, p[2], fmaf(w5, p[1], w4 * p[0])));
    const float2 b = base[i];
    float2 o = {fmaf(cn, vx, cb * b.x), fmaf(cn, vy, cb * b.y)};
    if (cy != 0.f) { const float2 yv = Y[i]; o.x = fmaf(cy, yv.x, o.x); o.y = fmaf(cy, yv.y, o.y); }
    if (cz != 0.f) { const float2 zv = z[i]; o.x = fmaf(cz, zv.x, o.x); o.y = fmaf(cz, zv.y, o.y); }
    dst[i] = o;
  }
}

// dst = a + cq * q  (prior sample x_T = Y + std(T) * z, sdes.py:197-202)
__global__ void caxpy_kernel(const float2* __restrict__ a, const float2* __restrict__ q, float cq, float2* __restrict__ dst, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float2 av = a[i], qv = q[i];
    float2 o = {fmaf(cq, qv.x, av.x), fmaf(cq, qv.y, av.y)};
    dst[i] = o;
  }
}

// ---- adaptive Dormand-Prince driver helpers (model.hip: fd_ode_solve_adaptive) ------------------------------------
struct fd_lincomb_args { const float2* k[7]; float c[7]; };
// dst = cx * x + dt * sum_i c[i] * k[i]   (null k[i] / zero c[i] are skipped by the host)
__global__ void ode_lincomb_kernel(const float2* __restrict__ x, float cx, float dt, fd_lincomb_args a, int nk, float2* __restrict__ dst, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float sx = 0.f, sy = 0.f;
    for (int j = 0; j < nk; ++j) { const float2 kv = a.k[j][i]; sx = fmaf(a.c[j], kv.x, sx); sy = fmaf(a.c[j], kv.y, sy); }
    float2 o = {dt * sx, dt * sy};
    if (cx != 0.f) { const float2 xv = x[i]; o.x = fmaf(cx, xv.x, o.x); o.y = fmaf(cx, xv.y, o.y); }
    dst[i] = o;
  }
}
// partial[block] = sum over the block's elements of |p - q|^2 / (atol + rtol * max(|r|, |s|))^2   (complex moduli; q may be null)
__global__ __launch_bounds__(256) void ode_scaled_sq_kernel(const float2* __restrict__ p, const float2* __restrict__ q, const float2* __restrict__ r,
                                                            const float2* __restrict__ s, float atol, float rtol, double* __restrict__ partial, long long n) {
  double acc = 0.0;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float2 d = p[i];
    if (q) { const float2 qv = q[i]; d.x -= qv.x; d.y -= qv.y; }
    const float2 rv = r[i], sv = s[i];
    const float m = fmaxf(sqrtf(rv.x * rv.x + rv.y * rv.y), sqrtf(sv.x * sv.x + sv.y * sv.y));
    const float sc = atol + rtol * m;
    acc += (double)((d.x * d.x + d.y * d.y) / (sc * sc));
  }
  acc = fd_wave_sum(acc);
  __shared__ double red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

inline int grid_for(long long n, int per_block = 256, int cap = 1 << 20) {
  long long g = (n + per_block - 1) / per_block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------
// C ABI + internal launchers
// ---------------------------------------------------------------------------------------------------------
extern "C" int fd_channel_sums_tiles(int H, int W) { return fd_cdiv((long long)H * W, 2048); }
int fd_combine_tiles(int H, int W) { return fd_cdiv((long long)H * W, COMBINE_PX); }

extern "C" int fd_channel_sums(const void* x, float* part, int B, int H, int W, int C, int dtype, void* stream) {
  FD_REQUIRE(x && part, "fd_channel_sums: null pointer");
  FD_REQUIRE(C >= 8 && C <= 256 && (C & (C - 1)) == 0, "fd_channel_sums: C must be a power of two in [8,256] (got %d)", C);
  FD_REQUIRE(dtype == FD_F32 || dtype == FD_BF16, "fd_channel_sums: bad dtype");
  hipStream_t st = fd_stream(stream);
  const int HW = H * W;
  const int ppb = 2048;  // pixels per block
  dim3 grid(fd_cdiv(HW, ppb), B);
  if (dtype == FD_BF16)
    hipLaunchKernelGGL(channel_sums_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)x, part, HW, C, ppb);
  else
    hipLaunchKernelGGL(channel_sums_kernel<float>, grid, dim3(256), 0, st, (const float*)x, part, HW, C, ppb);
  FD_LAUNCH_CHECK();
  return FD_OK;
}

extern "C" int fd_gn_silu_apply(const void* x, const float* affine, void* out, int B, long long hw, int C, int dtype, void* stream) {
  FD_REQUIRE(x && affine && out, "fd_gn_silu_apply: null pointer");
  FD_REQUIRE(B > 0 && hw > 0 && C > 0 && C % 8 == 0, "fd_gn_silu_apply: C must be a multiple of 8");
  FD_REQUIRE(dtype == FD_F32 || dtype == FD_BF16, "fd_gn_silu_apply: bad dtype");
  const long long nvec = (long long)B * hw * (C / 8);
  const dim3 grid(grid_for(nvec, 256, 1 << 16));
  if (dtype == FD_BF16) hipLaunchKernelGGL(gn_silu_apply_kernel<bf16>, grid, dim3(256), 0, fd_stream(stream), (const bf16*)x, affine, (bf16*)out, hw, C, nvec);
  else hipLaunchKernelGGL(gn_silu_apply_kernel<float>, grid, dim3(256), 0, fd_stream(stream), (const float*)x, affine, (float*)out, hw, C, nvec);
  FD_LAUNCH_CHECK();
  return FD_OK;
}

extern "C" int fd_gn_finalize(const float* part0, int tiles0, int stride0, int C0, const float* part1, int tiles1, int stride1,
                              int C1, const float* gamma, const float* beta, float* affine, int B, int groups, long long hw,
                              float eps, void* stream) {
  FD_REQUIRE(part0 && gamma && beta && affine, "fd_gn_finalize: null pointer");
  FD_REQUIRE((C1 == 0) == (part1 == nullptr), "fd_gn_finalize: part1 / C1 mismatch");
  FD_REQUIRE(tiles0 > 0 && stride0 >= C0 && (C1 == 0 || (tiles1 > 0 && stride1 >= C1)), "fd_gn_finalize: bad partial geometry");
  const int C = C0 + C1;
  FD_REQUIRE(groups > 0 && C % groups == 0 && C / groups <= 256, "fd_gn_finalize: C=%d not divisible by groups=%d", C, groups);
  const double inv_n = 1.0 / ((double)hw * (C / groups));
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(groups, B), dim3(256), 0, fd_stream(stream), part0, tiles0, stride0, C0, part1, tiles1,
                     stride1, C1, gamma, beta, affine, groups, inv_n, eps);
  FD_LAUNCH_CHECK();
  return FD_OK;
}

template <typename T, int VEC>
static int launch_fir(const void* x, const float* affine, void* out_raw, void* out_act, int B, int H, int W, int C,
                      int direction, hipStream_t st) {
  // Rows / output block per thread of the fused (activated) variants: big blocks re-use activated inputs (fewer SiLU evaluations per
  // output), small ones make more, shorter threads -- a small image is latency-bound (one clip at 384 x 64: 13-25 us with the big
  // blocks).  Every output is the same fma sequence in every variant, so the choice may depend on the batch size.
  auto blocks = [&](int rows, int cols, int n, int bx = 1) { return dim3(fd_cdiv((long long)B * fd_cdiv(rows, n) * fd_cdiv(cols, bx) * (C / VEC), 256)); };
  constexpr unsigned ENOUGH = 512;   // workgroups that keep 256 CUs busy
  if (direction > 0) {
#define FD_FIR_UP(ACT_, N_, BX_) hipLaunchKernelGGL((fir_up_kernel<T, VEC, ACT_, N_, BX_>), blocks(H, W, N_, BX_), dim3(256), 0, st, (const T*)x, affine, (T*)out_raw, (T*)out_act, B, H, W, C)
    if (!affine) FD_FIR_UP(false, 1, 1);
    // measured at 8 x 384 x 128 x 256 -> 768 x 256 (1.8 GB moved): 8 rows x 1 column, 8-channel vectors 414 us; 16 rows 446; 4 rows 432;
    // 4-channel vectors 409 / 458 / 463; two columns per thread (2.5 instead of 3.75 activations per input) 403 (4 ch x 8 rows): the up
    // direction does not react to the activation count or the vector width (4.4 TB/s, 4.0 of them stores) -- BX stays 1.  What it reacts
    // to is the store accounting (FAST): 357-361 us; FAST variants of the other shapes: 4 rows 404, 16 rows 392, 4-channel vectors 380-448
    else if (blocks(H, W, 8).x >= ENOUGH && H % 8 == 0 && out_raw && out_act)
      hipLaunchKernelGGL((fir_up_kernel<T, VEC, true, 8, 1, true>), blocks(H, W, 8, 1), dim3(256), 0, st, (const T*)x, affine, (T*)out_raw, (T*)out_act, B, H, W, C);
    else if (blocks(H, W, 8).x >= ENOUGH) FD_FIR_UP(true, 8, 1);
    else if (blocks(H, W, 2).x >= ENOUGH) FD_FIR_UP(true, 2, 1);
    else FD_FIR_UP(true, 1, 1);
#undef FD_FIR_UP
  } else {
    auto dgrid = [&](int by, int bx) { return dim3(fd_cdiv((long long)B * fd_cdiv(H / 2, by) * fd_cdiv(W / 2, bx) * (C / VEC), 256)); };
#define FD_FIR_DOWN(ACT_, BY_, BX_) hipLaunchKernelGGL((fir_down_kernel<T, VEC, ACT_, BY_, BX_>), dgrid(BY_, BX_), dim3(256), 0, st, (const T*)x, affine, (T*)out_raw, (T*)out_act, B, H, W, C)
    // measured on MI355X at 8 x 768 x 256 x 256 (bf16, dual output): 1x1 756 us, 2x2 675, 2x1 614, 8x1 565, 4x2 543, 4x1 467
    // round 2 (profiles/r02_fir_down_variants.txt): with 4-channel vectors the 4x2 block (6.25 instead of 10 SiLU evaluations per
    // output) fits the register budget: 430 us vs 512 us for 8-channel vectors x 4x1 at the same shape
    if (!affine) FD_FIR_DOWN(false, 1, 1);
    else if (VEC == 4) {
      // round 3: marching strips of NR output rows x 4 columns (1.33 instead of 1.875 activations per input, packed FIR arithmetic)
      auto mgrid = [&](int nr) { return dim3(fd_cdiv((long long)B * fd_cdiv(H / 2, nr) * fd_cdiv(W / 2, 4) * (C / VEC), 256)); };
#define FD_FIR_MARCH(NR_)                                                                                                                             \
  do {                                                                                                                                             \
    if (out_raw && (H / 2) % NR_ == 0 && (W / 2) % 4 == 0)                                                                                         \
      hipLaunchKernelGGL((fir_down_march_kernel<T, 4, NR_, true>), mgrid(NR_), dim3(256), 0, st, (const T*)x, affine, (T*)out_raw, (T*)out_act, B, H, W, C); \
    else                                                                                                                                           \
      hipLaunchKernelGGL((fir_down_march_kernel<T, 4, NR_, false>), mgrid(NR_), dim3(256), 0, st, (const T*)x, affine, (T*)out_raw, (T*)out_act, B, H, W, C); \
  } while (0)
      // measured at B = 8 x 256 channels (scripts/fir_bench.py): 768 x 256: 4x2 blocks 444 us, strips of 4 / 8 / 16 rows 315 / 302 / 292 us;
      // 384 x 128: 130 us, 102 / 97 / 120 us (16-row strips leave 1.5 workgroups per CU) -> the tallest strip with >= 3 workgroups per CU
      constexpr unsigned FILL = 768;
      if (out_act && mgrid(16).x >= FILL) FD_FIR_MARCH(16);
      else if (out_act && mgrid(8).x >= FILL) FD_FIR_MARCH(8);
      else if (out_act && mgrid(4).x >= FILL) FD_FIR_MARCH(4);
      else if (sizeof(T) == 2 && dgrid(4, 2).x >= ENOUGH) FD_FIR_DOWN(true, 4, 2);
      else if (sizeof(T) == 2 && dgrid(2, 1).x >= ENOUGH) FD_FIR_DOWN(true, 2, 1);
      else if (sizeof(T) == 4 && dgrid(4, 1).x >= ENOUGH) FD_FIR_DOWN(true, 4, 1);
      else FD_FIR_DOWN(true, 1, 1);
    } else if (dgrid(4, 1).x >= ENOUGH) FD_FIR_DOWN(true, 4, 1);
    else FD_FIR_DOWN(true, 1, 1);
#undef FD_FIR_DOWN
#undef FD_FIR_MARCH
  }
  FD_LAUNCH_CHECK();
  return FD_OK;
}

extern "C" int fd_fir_resample(const void* x, const float* affine, void* out_raw, void* out_act, int B, int H, int W,
                               int C, int direction, int dtype, void* stream) {
  FD_REQUIRE(x && (out_raw || out_act), "fd_fir_resample: null pointer");
  FD_REQUIRE(direction == 1 || direction == -1, "fd_fir_resample: direction must be +1 (up) or -1 (down)");
  FD_REQUIRE(C % 4 == 0, "fd_fir_resample: C must be a multiple of 4");
  FD_REQUIRE(direction > 0 || (H % 2 == 0 && W % 2 == 0), "fd_fir_resample: down needs even H, W");
  FD_REQUIRE(out_act == nullptr || affine != nullptr, "fd_fir_resample: out_act needs affine");
  hipStream_t st = fd_stream(stream);
  if (dtype == FD_BF16) {
    if (direction < 0 && affine) return launch_fir<bf16, 4>(x, affine, out_raw, out_act, B, H, W, C, direction, st);   // fused down: strips / 4x2 blocks
    if (C % 8 == 0) return launch_fir<bf16, 8>(x, affine, out_raw, out_act, B, H, W, C, direction, st);
    return launch_fir<bf16, 4>(x, affine, out_raw, out_act, B, H, W, C, direction, st);
  } else if (dtype == FD_F32) {
    return launch_fir<float, 4>(x, affine, out_raw, out_act, B, H, W, C, direction, st);
  }
  return fd_set_error(FD_EINVAL, "fd_fir_resample: bad dtype %d", dtype);
}

extern "C" int fd_upfirdn2d_out_size(int in_size, int up, int down, int pad0, int pad1, int ksize) {
  return (in_size * up + pad0 + pad1 - ksize + down) / down;  // upfirdn2d_kernel.cu:248-251
}

extern "C" int fd_upfirdn2d(const void* input, const float* kernel, void* out, int major, int in_h, int in_w, int minor,
                            int kernel_h, int kernel_w, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1,
                            int pad_y0, int pad_y1, int dtype, void* stream) {
  FD_REQUIRE(input && kernel && out, "fd_upfirdn2d: null pointer");
  FD_REQUIRE(up_x >= 1 && up_y >= 1 && down_x >= 1 && down_y >= 1, "fd_upfirdn2d: up/down factors must be >= 1");
  FD_REQUIRE(major > 0 && in_h > 0 && in_w > 0 && minor > 0 && kernel_h > 0 && kernel_w > 0, "fd_upfirdn2d: bad shape");
  const int out_h = fd_upfirdn2d_out_size(in_h, up_y, down_y, pad_y0, pad_y1, kernel_h);
  const int out_w = fd_upfirdn2d_out_size(in_w, up_x, down_x, pad_x0, pad_x1, kernel_w);
  FD_REQUIRE(out_h > 0 && out_w > 0, "fd_upfirdn2d: empty output (%d x %d)", out_h, out_w);
  const long long n = (long long)major * out_h * out_w * minor;
  dim3 grid(grid_for(n, 256, 65536));
  if (dtype == FD_F32)
    hipLaunchKernelGGL(upfirdn2d_kernel<float>, grid, dim3(256), 0, fd_stream(stream), (const float*)input, kernel, (float*)out, major,
                       in_h, in_w, minor, kernel_h, kernel_w, up_x, up_y, down_x, down_y, pad_x0, pad_y0, out_h, out_w);
  else if (dtype == FD_BF16)
    hipLaunchKernelGGL(upfirdn2d_kernel<bf16>, grid, dim3(256), 0, fd_stream(stream), (const bf16*)input, kernel, (bf16*)out, major,
                       in_h, in_w, minor, kernel_h, kernel_w, up_x, up_y, down_x, down_y, pad_x0, pad_y0, out_h, out_w);
  else
    return fd_set_error(FD_EINVAL, "fd_upfirdn2d: bad dtype %d", dtype);
  FD_LAUNCH_CHECK();
  return FD_OK;
}

extern "C" int fd_fused_bias_act(const float* x, const float* bias, float* out, long long n, int step_b, int size_b,
                                 int act, float alpha, float scale, void* stream) {
  FD_REQUIRE(x && out && n >= 0, "fd_fused_bias_act: bad arguments");
  FD_REQUIRE(act == 1 || act == 3, "fd_fused_bias_act: act must be 1 (linear) or 3 (lrelu)");
  FD_REQUIRE(bias == nullptr || (step_b > 0 && size_b > 0), "fd_fused_bias_act: bad bias geometry");
  if (n == 0) return FD_OK;
  hipLaunchKernelGGL(fused_bias_act_kernel, dim3(grid_for(n, 256, 65536)), dim3(256), 0, fd_stream(stream), x, bias, out, n,
                     step_b, size_b, act, alpha, scale);
  FD_LAUNCH_CHECK();
  return FD_OK;
}

int fd_time_embedding_impl(const float* t, float t_imm, int nt, const float* gfp_w, int nf, const float* w1, const float* b1,
                           const float* w2, const float* b2, float* temb, hipStream_t st) {
  hipLaunchKernelGGL(time_embedding_kernel, dim3(nt), dim3(256), sizeof(float) * 6 * nf, st, t, t_imm, gfp_w, nf, w1, b1, w2, b2, temb);
  FD_LAUNCH_CHECK();
  return FD_OK;
}

extern "C" int fd_time_embedding(const float* t, int nt, const float* gfp_w, int nf, const float* w1, const float* b1,
                                 const float* w2, const float* b2, float* temb, void* stream) {
  FD_REQUIRE(t && gfp_w && w1 && b1 && w2 && b2 && temb && nt > 0 && nf > 0, "fd_time_embedding: bad arguments");
  return fd_time_embedding_impl(t, 0.f, nt, gfp_w, nf, w1, b1, w2, b2, temb, fd_stream(stream));
}

int fd_temb_bias_batched(const fd_temb_job* jobs_dev, int njobs, const float* temb, int nt, int temb_dim, hipStream_t st) {
  hipLaunchKernelGGL(temb_bias_kernel, dim3(njobs, nt), dim3(256), sizeof(float) * temb_dim, st, jobs_dev, temb, temb_dim);
  FD_LAUNCH_CHECK();
  return FD_OK;
}

extern "C" int fd_temb_bias(const float* temb, int nt, int temb_dim, const float* dense_w, const float* dense_b,
                            const float* conv_bias, int Cout, float* out, void* stream) {
  FD_REQUIRE(temb && dense_w && dense_b && out && nt > 0 && temb_dim > 0 && Cout > 0, "fd_temb_bias: bad arguments");
  fd_temb_job j{dense_w, dense_b, conv_bias, out, Cout};
  hipLaunchKernelGGL(temb_bias_single_kernel, dim3(nt), dim3(256), sizeof(float) * temb_dim, fd_stream(stream), j, temb, temb_dim);
  FD_LAUNCH_CHECK();
  return FD_OK;
}

template <typename T>
static int edge_launch(int which, const fd_edge_args& a, hipStream_t st) {
  switch (which) {
    case 0: {  // pack input
      const long long n = (long long)a.B * a.H * a.W;
      hipLaunchKernelGGL(pack_input_kernel<T>, dim3(grid_for(n, 256, 8192)), dim3(256), 0, st, (const float2*)a.x, (const float2*)a.y, (T*)a.out, n);
      break;
    }
    case 2: {  // combine
      hipLaunchKernelGGL(combine_kernel<T>, dim3(fd_combine_tiles(a.H, a.W), a.B), dim3(256), 0, st, (const T*)a.x, a.w, a.bias, (const T*)a.y, (T*)a.out,
                         a.stats, a.H * a.W, a.Cout);
      break;
    }
    case 3: {  // output + update
      const long long n = (long long)a.B * a.H * a.W;
      hipLaunchKernelGGL(output_update_kernel<T>, dim3(grid_for(n, 256, 8192)), dim3(256), 0, st, (const T*)a.x, a.w, (const float2*)a.base, (const float2*)a.kold, a.coef, (float2*)a.out, (float2*)a.ksave, n);
      break;
    }
    case 4: {  // output + score-sampler update
      const long long n = (long long)a.B * a.H * a.W;
      hipLaunchKernelGGL(score_update_kernel<T>, dim3(grid_for(n, 256, 8192)), dim3(256), 0, st, (const T*)a.x, a.w, (const float2*)a.base, a.cb,
                         (const float2*)a.y, a.cy, a.coef, (const float2*)a.z, a.cz, (float2*)a.out, n);
      break;
    }
    case 5: {  // input convolution 4 -> Cout with GroupNorm partials (16 x 16 tiles)
      const dim3 grid((a.H >> 4) * (a.W >> 4), a.B);
#define FD_CONV_IN(NG_) hipLaunchKernelGGL((conv_in_kernel<T, NG_>), grid, dim3(256), 0, st, (const T*)a.x, a.w, a.bias, (T*)a.out, a.stats, a.H, a.W)
      switch (a.Cout) {
        case 8: FD_CONV_IN(1); break;
        case 16: FD_CONV_IN(2); break;
        case 32: FD_CONV_IN(4); break;
        case 64: FD_CONV_IN(8); break;
        default: return fd_set_error(FD_EINVAL, "edge_launch: conv_in needs Cout in {8, 16, 32, 64}");
      }
#undef FD_CONV_IN
      break;
    }
    default: return fd_set_error(FD_EINVAL, "edge_launch: bad op");
  }
  FD_LAUNCH_CHECK();
  return FD_OK;
}

int fd_edge_op(int which, const fd_edge_args& a, int dtype, hipStream_t st) {
  if (which == 5) FD_REQUIRE(a.x && a.w && a.bias && a.out && a.stats && a.H % 16 == 0 && a.W % 16 == 0, "edge op: conv_in needs whole 16 x 16 tiles, weights, bias and a stats buffer");
  if (which == 2) FD_REQUIRE(a.stats && a.Cout >= 8 && a.Cout <= 256 && (a.Cout & (a.Cout - 1)) == 0, "edge op: combine needs a stats buffer and a power-of-two Cout in [8, 256]");
  return dtype == FD_BF16 ? edge_launch<bf16>(which, a, st) : edge_launch<float>(which, a, st);
}

extern "C" int fd_conv_in(const void* in8, const float* w, const float* bias, void* out, float* stats, int B, int H, int W, int Cout, int dtype,
                          void* stream) {
  FD_REQUIRE(dtype == FD_BF16 || dtype == FD_F32, "fd_conv_in: dtype must be FD_BF16 or FD_F32");
  FD_REQUIRE(B > 0 && H > 0 && W > 0, "fd_conv_in: bad shape");
  fd_edge_args a;
  a.x = in8; a.w = w; a.bias = bias; a.out = out; a.stats = stats; a.B = B; a.H = H; a.W = W; a.Cout = Cout;
  return fd_edge_op(5, a, dtype, fd_stream(stream));
}

int fd_init_state(const float* Y, const float* noise, const double* sigma_dev, int sigma_n, float sigma_fac, float* x0, int B,
                  int F, int T, hipStream_t st) {
  const long long n = (long long)B * F * T;
  hipLaunchKernelGGL(init_state_kernel, dim3(grid_for(n, 256, 8192)), dim3(256), 0, st, (const float2*)Y, (const float2*)noise, sigma_dev, sigma_n,
                     sigma_fac, (float2*)x0, F, T, n);
  FD_LAUNCH_CHECK();
  return FD_OK;
}

int fd_caxpy(const float* a, const float* q, float cq, float* dst, long long n, hipStream_t st) {
  hipLaunchKernelGGL(caxpy_kernel, dim3(grid_for(n, 256, 8192)), dim3(256), 0, st, (const float2*)a, (const float2*)q, cq, (float2*)dst, n);
  FD_LAUNCH_CHECK();
  return FD_OK;
}

int fd_ode_lincomb(const float* x, float cx, float dt, const float* const* k, const float* c, int nk, float* dst, long long n, hipStream_t st) {
  fd_lincomb_args a;
  int m = 0;
  for (int j = 0; j < nk && j < 7; ++j)
    if (k[j] && c[j] != 0.f) { a.k[m] = (const float2*)k[j]; a.c[m] = c[j]; ++m; }
  for (int j = m; j < 7; ++j) { a.k[j] = nullptr; a.c[j] = 0.f; }
  hipLaunchKernelGGL(ode_lincomb_kernel, dim3(grid_for(n, 256, 8192)), dim3(256), 0, st, (const float2*)x, cx, dt, a, m, (float2*)dst, n);
  FD_LAUNCH_CHECK();
  return FD_OK;
}

int fd_ode_scaled_sq(const float* p, const float* q, const float* r, const float* s, float atol, float rtol, double* partial, int nblocks,
                     long long n, hipStream_t st) {
  hipLaunchKernelGGL(ode_scaled_sq_kernel, dim3(nblocks), dim3(256), 0, st, (const float2*)p, (const float2*)q, (const float2*)r, (const float2*)s,
                     atol, rtol, partial, n);
  FD_LAUNCH_CHECK();
  return FD_OK;
}
