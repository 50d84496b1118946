// ndac.hip -- the NDAC codec in front of the FlowDec post-filter (demo.ipynb cells 2-3: `DAC.load`, `dac_model.encode(x,
// n_quantizers=nq)`, `dac_model.quantizer.from_codes(codes)`, `dac_model.decode(zq)`): the Descript Audio Codec architecture
// (descript-audio-codec==1.0.0, /root/reference/requirements.txt:4 -- third party, NOT under /root/reference: the algorithm below
// is the published one, restated; oracle/ndac_oracle.py names the upstream definitions; PARITY UNPINNED).
//
//   encoder  WNConv1d(1, d, 7) -> [ 3 x ResidualUnit(dilation 1, 3, 9) -> Snake -> WNConv1d(k = 2 s, stride s) ] per rate -> Snake -> WNConv1d(., latent, 3)
//   RVQ      per codebook: in_proj (1x1) -> L2-normalised nearest neighbour over the codebook -> out_proj (1x1); residual chain
//   decoder  WNConv1d(latent, D, 7) -> [ Snake -> WNConvTranspose1d(k = 2 s, stride s) -> 3 x ResidualUnit ] per rate -> Snake -> WNConv1d(., 1, 7) -> tanh
//
// Layout [B][C][T] float32 (the reference's Conv1d layout).  This file is the EXACT path: f32 on the vector ALUs in a defined operation
// order per output (input channels ascending, taps ascending, one fma each; every convolution kernel below produces the same bits),
// which is what makes the encoder's code indices reproducible.  The wide layers also exist as split-bf16 implicit GEMMs on the matrix
// cores (ndac_mfma.hip: the decoder's default, the encoder's opt-in -- fd_ndac_set_precision).  Kernels:
//   conv1d_kernel       LDS-tiled dilated / strided 1-D convolution, 128 outputs x 32 channels per workgroup; optional Snake of the INPUT
//                       while the tile is staged (operator-level ABI), bias / residual add (ResidualUnit: x + block(x)) / tanh and the
//                       NEXT layer's Snake (second output) in the epilogue
//   conv1d_s1_kernel    the stride-1 layers (K = 7 dilation 1 / 3 / 9, K = 1, K = 3): register windows of 4 consecutive outputs
//   conv1d_ci1_kernel / conv1d_co1_kernel   the one-channel ends of the codec (audio -> d channels, d channels -> audio)
//   convtr1d_kernel     transposed convolution as a gather: output n takes taps k = (n + p) mod s, + s, ... of inputs (n + p - k) / s
//   rvq_step_kernel     one residual quantiser: in_proj with f64 accumulation, nearest neighbour with the float32 operation order the
//                       oracle defines (code indices BIT-EXACT; ties -> lowest index), straight-through value, out_proj, residual update
//   rvq_from_codes_kernel   codes -> sum_i out_proj_i(codebook_i[code])
#include <math.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "internal.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int TT = 128;   // output positions per workgroup
constexpr int CT = 32;    // output channels per workgroup
constexpr int CIC = 8;    // input channels per LDS chunk
constexpr int CTP = CT + 4;   // padded weight row of the transposed convolution

__device__ __forceinline__ float snake_f(float x, float alpha) {
  // x + (alpha + 1e-9)^-1 sin^2(alpha x)   (dac/nn/layers.py `snake`); IEEE division, sinf at full precision
  const float s = sinf(alpha * x);
  return x + (1.0f / (alpha + 1e-9f)) * (s * s);
}

struct Conv1dArgs {
  const float* x; const float* w; const float* bias; const float* alpha; const float* res; float* out;
  float* out_act; const float* alpha_out;   // optional second output: snake(result, alpha_out[co]) -- the NEXT layer's activated input,
                                            // computed once by the producer instead of once per consuming workgroup (24x at 768 channels)
  int B, Ci, T, Co, K, To, stride, pad, dil, tanh_out;
  int wt;   // weight layout: 0 = [Co][Ci][K] (nn.Conv1d, the operator-level ABI), 1 = [Ci][K][Co] (the codec's own copy: coalesced staging)
};

// weights are read as [Co][Ci][K] (PyTorch layout) and staged as [ci][k][co] so that a thread's 4 output channels are one b128 read
__global__ __launch_bounds__(256) void conv1d_kernel(Conv1dArgs a) {
  extern __shared__ float sm[];
  const int span = (TT - 1) * a.stride + (a.K - 1) * a.dil + 1;
  float* xs = sm;                         // [CIC][span]
  float* ws = sm + CIC * span;            // [CIC][K][CT]
  const int b = blockIdx.z, co0 = blockIdx.y * CT, t0 = blockIdx.x * TT;
  const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;   // thread: outputs t0 + tx + 32 j (j < 4) x channels co0 + 4 ty .. + 3
  f32x2 acc[4][2];   // [output j][channel pair]: v_pk_fma_f32, two channels per instruction (the same IEEE fma per element)
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j][0] = acc[j][1] = f32x2{0.f, 0.f};
  const long long in0 = (long long)t0 * a.stride - a.pad;   // input position of xs[.][0]
  for (int c0 = 0; c0 < a.Ci; c0 += CIC) {
    __syncthreads();
    for (int i = tid; i < CIC * span; i += 256) {
      const int ci = i / span, p = i - ci * span;
      const long long tin = in0 + p;
      float v = 0.f;
      if (c0 + ci < a.Ci && tin >= 0 && tin < a.T) {
        v = a.x[((size_t)b * a.Ci + c0 + ci) * a.T + tin];
        if (a.alpha) v = snake_f(v, a.alpha[c0 + ci]);      // zero padding applies to the ACTIVATED signal (Snake, then Conv1d(padding=..))
      }
      xs[i] = v;
    }
    for (int i = tid; i < CIC * a.K * CT; i += 256) {
      const int co = i % CT, r = i / CT, k = r % a.K, ci = r / a.K;
      ws[i] = (c0 + ci < a.Ci && co0 + co < a.Co) ? a.w[a.wt ? ((size_t)(c0 + ci) * a.K + k) * a.Co + co0 + co : ((size_t)(co0 + co) * a.Ci + c0 + ci) * a.K + k] : 0.f;
    }
    __syncthreads();
#pragma unroll 1
    for (int ci = 0; ci < CIC; ++ci) {
      const float* xr = xs + ci * span + tx * a.stride;
      const float* wr = ws + ci * a.K * CT + ty * 4;
      for (int k = 0; k < a.K; ++k) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + k * CT);
        const f32x2 w01 = {wv[0], wv[1]}, w23 = {wv[2], wv[3]};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float xv = xr[32 * j * a.stride + k * a.dil];
          const f32x2 x2 = {xv, xv};
          acc[j][0] = __builtin_elementwise_fma(x2, w01, acc[j][0]);
          acc[j][1] = __builtin_elementwise_fma(x2, w23, acc[j][1]);
        }
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int co = co0 + ty * 4 + c;
    if (co >= a.Co) continue;
    const float bv = a.bias ? a.bias[co] : 0.f;
    const float ao = a.out_act ? a.alpha_out[co] : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int t = t0 + tx + 32 * j;
      if (t >= a.To) continue;
      const size_t o = ((size_t)b * a.Co + co) * a.To + t;
      float v = acc[j][c >> 1][c & 1] + bv;
      if (a.res) v += a.res[o];
      if (a.tanh_out) v = tanhf(v);
      if (a.out) a.out[o] = v;
      if (a.out_act) a.out_act[o] = snake_f(v, ao);
    }
  }
}

// Stride-1 convolutions with a compile-time tap count and dilation (every ResidualUnit conv: K = 7 with dilation 1 / 3, K = 1; the
// K = 3 latent conv): a thread owns two groups of 4 CONSECUTIVE outputs (t0 + 4 tx + {0..3} and 128 further) x 4 channels.  Per input
// channel the 4 + (K - 1) DIL inputs a group needs are read ONCE as aligned ds_read_b128 (lane stride 16 B: conflict-free) into a
// register window and all K taps run out of registers -- 13 wide LDS reads per 224 packed-FMA pairs (K = 7, DIL = 1) where the generic
// kernel above issues one ds_read_b32 per 4 FMAs and is bound by the LDS pipe.  256 outputs x 32 channels per workgroup.
// (8 channels per thread -- 64 per workgroup, twice the FMAs per weight read -- measured SLOWER: encode 26.7 -> 31.1 ms at 146-166
// registers, three waves per SIMD instead of five.)
constexpr int TT1 = 256;
template <int K, int DIL>
__global__ __launch_bounds__(256) void conv1d_s1_kernel(Conv1dArgs a) {
  extern __shared__ float sm[];
  constexpr int HALO = (K - 1) * DIL;
  constexpr int SPAN = (TT1 + HALO + 3) / 4 * 4;     // floats per staged input row (multiple of 4: aligned b128 rows)
  constexpr int NW = (4 + HALO + 3) / 4;             // b128 reads per group window
  float* xs = sm;                                    // [CIC][SPAN]
  float* ws = sm + CIC * SPAN;                       // [CIC][K][CT]
  const int b = blockIdx.z, co0 = blockIdx.y * CT, t0 = blockIdx.x * TT1;
  const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
  f32x2 acc[2][4][2];   // [group][output][channel pair]
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[g][i][0] = acc[g][i][1] = f32x2{0.f, 0.f};
  const long long in0 = (long long)t0 - a.pad;
  for (int c0 = 0; c0 < a.Ci; c0 += CIC) {
    __syncthreads();
    for (int i = tid; i < CIC * SPAN; i += 256) {
      const int ci = i / SPAN, p = i - ci * SPAN;
      const long long tin = in0 + p;
      float v = 0.f;
      if (c0 + ci < a.Ci && tin >= 0 && tin < a.T) {
        v = a.x[((size_t)b * a.Ci + c0 + ci) * a.T + tin];
        if (a.alpha) v = snake_f(v, a.alpha[c0 + ci]);
      }
      xs[i] = v;
    }
    for (int i = tid; i < CIC * K * CT; i += 256) {
      const int co = i % CT, r = i / CT, k = r % K, ci = r / K;
      ws[i] = (c0 + ci < a.Ci && co0 + co < a.Co) ? a.w[a.wt ? ((size_t)(c0 + ci) * K + k) * a.Co + co0 + co : ((size_t)(co0 + co) * a.Ci + c0 + ci) * K + k] : 0.f;
    }
    __syncthreads();
#pragma unroll 1
    for (int ci = 0; ci < CIC; ++ci) {
      const float* wr = ws + ci * K * CT + ty * 4;
      f32x2 w01[K], w23[K];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + k * CT);
        w01[k] = f32x2{wv[0], wv[1]}; w23[k] = f32x2{wv[2], wv[3]};
      }
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const float* xr = xs + ci * SPAN + 4 * tx + 128 * g;
        float xw[4 * NW];
#pragma unroll
        for (int m = 0; m < NW; ++m) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(xr + 4 * m);
          xw[4 * m] = v[0]; xw[4 * m + 1] = v[1]; xw[4 * m + 2] = v[2]; xw[4 * m + 3] = v[3];
        }
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const f32x2 x2 = {xw[i + k * DIL], xw[i + k * DIL]};
            acc[g][i][0] = __builtin_elementwise_fma(x2, w01[k], acc[g][i][0]);
            acc[g][i][1] = __builtin_elementwise_fma(x2, w23[k], acc[g][i][1]);
          }
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int co = co0 + ty * 4 + c;
    if (co >= a.Co) continue;
    const float bv = a.bias ? a.bias[co] : 0.f;
    const float ao = a.out_act ? a.alpha_out[co] : 0.f;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int t = t0 + 4 * tx + 128 * g + i;
        if (t >= a.To) continue;
        const size_t o = ((size_t)b * a.Co + co) * a.To + t;
        float v = acc[g][i][c >> 1][c & 1] + bv;
        if (a.res) v += a.res[o];
        if (a.tanh_out) v = tanhf(v);
        if (a.out) a.out[o] = v;
        if (a.out_act) a.out_act[o] = snake_f(v, ao);
      }
  }
}

// The two ends of the codec, where the 32-channel tiles above are 97 % padding: ONE input channel (the encoder's first layer:
// audio -> d channels) and ONE output channel (the decoder's last layer: d channels -> audio, tanh).  Same fma sequence per output
// as conv1d_kernel (input channels ascending, taps ascending, then + bias): identical bits.
__global__ __launch_bounds__(256) void conv1d_ci1_kernel(Conv1dArgs a) {   // Ci == 1, stride 1, K <= 8, no input activation
  const int b = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
  if (t >= a.To) return;
  float xv[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const long long tin = (long long)t - a.pad + (long long)k * a.dil;
    xv[k] = k < a.K && tin >= 0 && tin < a.T ? a.x[(size_t)b * a.T + tin] : 0.f;
  }
  for (int co = 0; co < a.Co; ++co) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k < a.K) acc = __builtin_fmaf(xv[k], a.w[a.wt ? k * a.Co + co : co * a.K + k], acc);
    const size_t o = ((size_t)b * a.Co + co) * a.To + t;
    float v = acc + (a.bias ? a.bias[co] : 0.f);
    if (a.res) v += a.res[o];
    if (a.tanh_out) v = tanhf(v);
    if (a.out) a.out[o] = v;
    if (a.out_act) a.out_act[o] = snake_f(v, a.alpha_out[co]);
  }
}

constexpr int C1C = 16;   // input channels per LDS chunk of the one-output-channel kernel
__global__ __launch_bounds__(256) void conv1d_co1_kernel(Conv1dArgs a) {   // Co == 1, stride 1
  extern __shared__ float sm[];
  const int span = 256 + (a.K - 1) * a.dil;
  const int b = blockIdx.y, t0 = blockIdx.x * 256, tid = threadIdx.x;
  const long long in0 = (long long)t0 - a.pad;
  float acc = 0.f;
  for (int c0 = 0; c0 < a.Ci; c0 += C1C) {
    __syncthreads();
    for (int i = tid; i < C1C * span; i += 256) {
      const int ci = i / span, p = i - ci * span;
      const long long tin = in0 + p;
      float v = 0.f;
      if (c0 + ci < a.Ci && tin >= 0 && tin < a.T) {
        v = a.x[((size_t)b * a.Ci + c0 + ci) * a.T + tin];
        if (a.alpha) v = snake_f(v, a.alpha[c0 + ci]);
      }
      sm[i] = v;
    }
    __syncthreads();
    const int cn = a.Ci - c0 < C1C ? a.Ci - c0 : C1C;
    for (int ci = 0; ci < cn; ++ci) {
      const float* xr = sm + ci * span + tid;
      const float* wr = a.w + (size_t)(c0 + ci) * a.K;   // [Co = 1][Ci][K] and [Ci][K][Co = 1] are the same array
      for (int k = 0; k < a.K; ++k) acc = __builtin_fmaf(xr[k * a.dil], wr[k], acc);
    }
  }
  const int t = t0 + tid;
  if (t >= a.To) return;
  const size_t o = (size_t)b * a.To + t;
  float v = acc + (a.bias ? a.bias[0] : 0.f);
  if (a.res) v += a.res[o];
  if (a.tanh_out) v = tanhf(v);
  if (a.out) a.out[o] = v;
  if (a.out_act) a.out_act[o] = snake_f(v, a.alpha_out[0]);
}

struct ConvTr1dArgs {
  const float* x; const float* w; const float* bias; const float* alpha; float* out;
  float* out_act; const float* alpha_out;
  int B, Ci, T, Co, K, To, stride, pad;
  int wt;   // 0 = [Ci][Co][K] (nn.ConvTranspose1d), 1 = [Ci][K][Co]
};

// out[b][co][n] = bias[co] + sum_ci sum_{k = (n + p) mod s, += s, < K} act(x)[b][ci][(n + p - k) / s] * w[ci][co][k]   (w: [Ci][Co][K])
__global__ __launch_bounds__(256) void convtr1d_kernel(ConvTr1dArgs a) {
  extern __shared__ float sm[];
  const int ntap = (a.K + a.stride - 1) / a.stride;
  const int tspan = TT / a.stride + ntap + 1;
  float* xs = sm;                         // [CIC][tspan]
  float* ws = sm + CIC * tspan;           // [CIC][K][CTP]: rows padded by 16 B -- the tap index k varies per LANE here (k = (n + p) mod s),
                                          // and rows 128 B apart would put the lanes of a b128 group on 2 of its 16 slots
  const int b = blockIdx.z, co0 = blockIdx.y * CT, n0 = blockIdx.x * TT;
  const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
  const int tbase = (n0 + a.pad) / a.stride - (ntap - 1);   // input position of xs[.][0] (may be negative)
  float acc[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[j][c] = 0.f;
  int kk0[4], tt0[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + tx + 32 * j;
    kk0[j] = (n + a.pad) % a.stride;
    tt0[j] = (n + a.pad) / a.stride - tbase;   // xs index of the tap k = kk0; tap k + m s reads index tt0 - m
  }
  for (int c0 = 0; c0 < a.Ci; c0 += CIC) {
    __syncthreads();
    for (int i = tid; i < CIC * tspan; i += 256) {
      const int ci = i / tspan, p = i - ci * tspan, tin = tbase + p;
      float v = 0.f;
      if (c0 + ci < a.Ci && tin >= 0 && tin < a.T) {
        v = a.x[((size_t)b * a.Ci + c0 + ci) * a.T + tin];
        if (a.alpha) v = snake_f(v, a.alpha[c0 + ci]);
      }
      xs[i] = v;
    }
    for (int i = tid; i < CIC * a.K * CT; i += 256) {
      const int co = i % CT, r = i / CT, k = r % a.K, ci = r / a.K;
      ws[r * CTP + co] = (c0 + ci < a.Ci && co0 + co < a.Co) ? a.w[a.wt ? ((size_t)(c0 + ci) * a.K + k) * a.Co + co0 + co : ((size_t)(c0 + ci) * a.Co + co0 + co) * a.K + k] : 0.f;
    }
    __syncthreads();
#pragma unroll 1
    for (int ci = 0; ci < CIC; ++ci) {
      const float* xr = xs + ci * tspan;
      const float* wr = ws + ci * a.K * CTP + ty * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        for (int m = 0, k = kk0[j]; k < a.K; ++m, k += a.stride) {
          const float xv = xr[tt0[j] - m];
          const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + k * CTP);
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[j][c] = fmaf(xv, wv[c], acc[j][c]);
        }
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int co = co0 + ty * 4 + c;
    if (co >= a.Co) continue;
    const float bv = a.bias ? a.bias[co] : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx + 32 * j;
      if (n < a.To) {
        const size_t o = ((size_t)b * a.Co + co) * a.To + n;
        const float v = acc[j][c] + bv;
        if (a.out) a.out[o] = v;
        if (a.out_act) a.out_act[o] = snake_f(v, a.alpha_out[co]);
      }
    }
  }
}

// ---- residual vector quantiser -------------------------------------------------------------------------------------------
constexpr int RT = 8;    // time steps per workgroup

struct RvqArgs {
  float* residual;        // [B][D][T] in / out
  float* zq;              // [B][D][T] accumulated
  int* codes;             // [B][nq][T], this quantiser writes row q
  float* latents;         // [B][nq * cd][T] or null
  const float* win; const float* bin;     // [cd][D], [cd]
  const float* wout; const float* bout;   // [D][cd], [D]
  const float* cb;        // [J][cd] raw codebook
  const float* cbn;       // [J][cd] L2-normalised rows (host, the oracle's float32 operation order)
  const float* c2;        // [J] sum of squares of the normalised rows
  int B, D, T, J, cd, q, nq;
};

// The float32 operation order of oracle/ndac_oracle.py `vq_nearest` / `l2_normalize_rows_f32` (no fused multiply-add anywhere).
// 8 time steps per workgroup (a 2 s clip has 150 frames: small tiles are what fills the chip); the 256 threads are
//   (1) 8 steps x 8 codebook dims x 4 quarters of the latent channels (f64 partial sums, combined in a fixed order),
//   (3) 8 steps x 32 slices of the codebook, each scanned in ascending order with a strict '<', slices merged in ascending order
//       (ties -> lowest index), (4) 8 steps x 32 output-channel lanes.
__global__ __launch_bounds__(256) void rvq_step_kernel(RvqArgs a) {
#pragma clang fp contract(off)
  constexpr int NP = 256 / RT;       // codebook slices / output-channel lanes per time step
  static_assert(NP == 32, "the in_proj split below is 8 dims x 4 channel quarters");
  __shared__ double pz[4][RT][8];
  __shared__ float ze[RT][9];        // in_proj output per time step (cd <= 8: fd_ndac_create), +1 pad
  __shared__ float en[RT][9];
  __shared__ float e2s[RT];
  __shared__ float bestd[NP][RT];
  __shared__ int besti[NP][RT];
  __shared__ float stv[RT][9];       // straight-through value z_e + (c - z_e)
  const int b = blockIdx.y, t0 = blockIdx.x * RT, tid = threadIdx.x;
  const int tt = tid % RT, part = tid / RT;
  const int t = t0 + tt;
  const bool tv = t < a.T;
  // (1) z_e[d] = sum_c win[d][c] * residual[c] + bin[d]: exact f32 products accumulated in f64, rounded to f32 once
  {
    const int d = part & 7, qd = part >> 3;   // NP = 32: 8 dims x 4 channel quarters
    double acc = 0.0;
    if (tv && d < a.cd) {
      const int c0 = (int)((long long)a.D * qd / 4), c1 = (int)((long long)a.D * (qd + 1) / 4);
      const float* rp = a.residual + (size_t)b * a.D * a.T + t;
      const float* wp = a.win + (size_t)d * a.D;
      for (int c = c0; c < c1; ++c) acc += (double)wp[c] * (double)rp[(size_t)c * a.T];
    }
    pz[qd][tt][d] = acc;
  }
  __syncthreads();
  if (part < 8) {
    const int d = part;
    ze[tt][d] = d < a.cd && tv ? (float)((((pz[0][tt][d] + pz[1][tt][d]) + pz[2][tt][d]) + pz[3][tt][d]) + (double)a.bin[d]) : 0.f;
  }
  __syncthreads();
  // (2) normalise: s = sum x^2 (sequential), n = sqrt(s), x / max(n, 1e-12); e2 = sum en^2
  if (part == 0) {
    float s = 0.f;
    for (int d = 0; d < a.cd; ++d) s = s + ze[tt][d] * ze[tt][d];
    const float n = fmaxf(sqrtf(s), 1e-12f);
    float e2 = 0.f;
    for (int d = 0; d < a.cd; ++d) {
      const float v = ze[tt][d] / n;
      en[tt][d] = v;
      e2 = e2 + v * v;
    }
    e2s[tt] = e2;
  }
  __syncthreads();
  // (3) nearest neighbour
  {
    float e[8];
    for (int d = 0; d < 8; ++d) e[d] = d < a.cd ? en[tt][d] : 0.f;
    const float e2 = e2s[tt];
    const int j0 = (int)((long long)a.J * part / NP), j1 = (int)((long long)a.J * (part + 1) / NP);
    float bd = INFINITY; int bi = j0;
    for (int j = j0; j < j1; ++j) {
      const float* c = a.cbn + (size_t)j * a.cd;
      float dot = 0.f;
      for (int d = 0; d < a.cd; ++d) dot = dot + e[d] * c[d];
      const float dist = (e2 - 2.0f * dot) + a.c2[j];
      if (dist < bd) { bd = dist; bi = j; }
    }
    bestd[part][tt] = bd; besti[part][tt] = bi;
  }
  __syncthreads();
  if (part == 0) {
    float bd = bestd[0][tt]; int bi = besti[0][tt];
    for (int p = 1; p < NP; ++p)
      if (bestd[p][tt] < bd) { bd = bestd[p][tt]; bi = besti[p][tt]; }
    if (tv) a.codes[((size_t)b * a.nq + a.q) * a.T + t] = bi;
    for (int d = 0; d < a.cd; ++d) {
      const float z = ze[tt][d];
      stv[tt][d] = z + (a.cb[(size_t)bi * a.cd + d] - z);     // z_e + (z_q - z_e).detach()
      if (tv && a.latents) a.latents[((size_t)b * a.nq * a.cd + (size_t)a.q * a.cd + d) * a.T + t] = z;
    }
  }
  __syncthreads();
  // (4) z_q_i[c] = sum_d wout[c][d] * st[d] + bout[c] (f64, rounded once);  zq += z_q_i;  residual -= z_q_i
  if (tv) {
    for (int c = part; c < a.D; c += NP) {
      double acc = 0.0;
      const float* wp = a.wout + (size_t)c * a.cd;
      for (int d = 0; d < a.cd; ++d) acc += (double)wp[d] * (double)stv[tt][d];
      const float zi = (float)(acc + (double)a.bout[c]);
      const size_t o = ((size_t)b * a.D + c) * a.T + t;
      a.zq[o] = a.zq[o] + zi;
      a.residual[o] = a.residual[o] - zi;
    }
  }
}

struct FromCodesArgs {
  const int* codes; float* zq; const float* const* wout; const float* const* bout; const float* const* cb;
  int B, D, T, J, cd, nq;
};
// z_q[b][c][t] = sum_i ( f32( sum_d wout_i[c][d] cb_i[code][d] + bout_i[c] ) ), i ascending, f32 adds  (ResidualVectorQuantize.from_codes)
__global__ __launch_bounds__(256) void rvq_from_codes_kernel(FromCodesArgs a) {
#pragma clang fp contract(off)
  const long long n = (long long)a.B * a.D * a.T;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int t = (int)(i % a.T);
    const int c = (int)((i / a.T) % a.D);
    const int b = (int)(i / ((long long)a.T * a.D));
    float z = 0.f;
    for (int q = 0; q < a.nq; ++q) {
      int code = a.codes[((size_t)b * a.nq + q) * a.T + t];
      code = code < 0 ? 0 : (code >= a.J ? a.J - 1 : code);
      const float* e = a.cb[q] + (size_t)code * a.cd;
      const float* wp = a.wout[q] + (size_t)c * a.cd;
      double acc = 0.0;
      for (int d = 0; d < a.cd; ++d) acc += (double)wp[d] * (double)e[d];
      z = z + (float)(acc + (double)a.bout[q][c]);
    }
    a.zq[i] = z;
  }
}

size_t conv1d_lds(int K, int stride, int dil) { return sizeof(float) * (size_t)(CIC * ((TT - 1) * stride + (K - 1) * dil + 1) + CIC * K * CT); }
size_t convtr_lds(int K, int stride) { return sizeof(float) * (size_t)(CIC * (TT / stride + (K + stride - 1) / stride + 1) + CIC * K * CTP); }

int launch_conv1d(const float* x, const float* w, const float* bias, const float* alpha, const float* res, float* out, int B, int Ci, int T, int Co,
                  int K, int stride, int pad, int dil, int tanh_out, hipStream_t st, int wt = 0, float* out_act = nullptr, const float* alpha_out = nullptr) {
  FD_REQUIRE(B > 0 && Ci > 0 && T > 0 && Co > 0 && K > 0 && stride > 0 && dil > 0 && pad >= 0, "fd_conv1d: bad shape");
  const long long To = ((long long)T + 2 * pad - (long long)dil * (K - 1) - 1) / stride + 1;
  FD_REQUIRE(To > 0 && To < (1ll << 31), "fd_conv1d: empty output");
  const size_t lds = conv1d_lds(K, stride, dil);
  FD_REQUIRE(lds <= 64 * 1024, "fd_conv1d: kernel %d / stride %d / dilation %d needs %zu bytes of LDS (limit 64 KiB)", K, stride, dil, lds);
  Conv1dArgs a{x, w, bias, alpha, res, out, out_act, alpha_out, B, Ci, T, Co, K, (int)To, stride, pad, dil, tanh_out, wt};
  // the accumulation order per output is the same in both kernels (input channels ascending, taps ascending, one fma each): identical bits
  if (stride == 1 && Ci == 1 && K <= 8 && !alpha) {
    hipLaunchKernelGGL(conv1d_ci1_kernel, dim3(fd_cdiv(To, 256), B), dim3(256), 0, st, a);
    FD_LAUNCH_CHECK();
    return FD_OK;
  }
  if (stride == 1 && Co == 1 && sizeof(float) * C1C * (256 + (size_t)(K - 1) * dil) <= 64 * 1024) {
    hipLaunchKernelGGL(conv1d_co1_kernel, dim3(fd_cdiv(To, 256), B), dim3(256), sizeof(float) * C1C * (256 + (size_t)(K - 1) * dil), st, a);
    FD_LAUNCH_CHECK();
    return FD_OK;
  }
#define FD_CONV1D_S1(K_, D_)                                                                                                                      \
  if (stride == 1 && K == K_ && dil == D_ && Co >= 4) {                                                                                           \
    constexpr size_t lds1 = sizeof(float) * (size_t)(CIC * ((TT1 + (K_ - 1) * D_ + 3) / 4 * 4) + CIC * K_ * CT);                                    \
    hipLaunchKernelGGL((conv1d_s1_kernel<K_, D_>), dim3(fd_cdiv(To, TT1), fd_cdiv(Co, CT), B), dim3(256), lds1, st, a);                            \
    FD_LAUNCH_CHECK();                                                                                                                            \
    return FD_OK;                                                                                                                                 \
  }
  FD_CONV1D_S1(7, 1) FD_CONV1D_S1(7, 3) FD_CONV1D_S1(7, 9) FD_CONV1D_S1(1, 1) FD_CONV1D_S1(3, 1)
#undef FD_CONV1D_S1
  hipLaunchKernelGGL(conv1d_kernel, dim3(fd_cdiv(To, TT), fd_cdiv(Co, CT), B), dim3(256), lds, st, a);
  FD_LAUNCH_CHECK();
  return FD_OK;
}

int launch_convtr1d(const float* x, const float* w, const float* bias, const float* alpha, float* out, int B, int Ci, int T, int Co, int K, int stride,
                    int pad, hipStream_t st, int wt = 0, float* out_act = nullptr, const float* alpha_out = nullptr) {
  FD_REQUIRE(B > 0 && Ci > 0 && T > 0 && Co > 0 && K > 0 && stride > 0 && pad >= 0, "fd_conv_transpose1d: bad shape");
  const long long To = ((long long)T - 1) * stride - 2 * pad + K;
  FD_REQUIRE(To > 0 && To < (1ll << 31), "fd_conv_transpose1d: empty output");
  const size_t lds = convtr_lds(K, stride);
  FD_REQUIRE(lds <= 64 * 1024, "fd_conv_transpose1d: kernel %d needs %zu bytes of LDS (limit 64 KiB)", K, lds);
  ConvTr1dArgs a{x, w, bias, alpha, out, out_act, alpha_out, B, Ci, T, Co, K, (int)To, stride, pad, wt};
  hipLaunchKernelGGL(convtr1d_kernel, dim3(fd_cdiv(To, TT), fd_cdiv(Co, CT), B), dim3(256), lds, st, a);
  FD_LAUNCH_CHECK();
  return FD_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// model
// ---------------------------------------------------------------------------------------------------------------------
struct Param { std::string name; std::vector<int> shape; long long numel() const { long long n = 1; for (int s : shape) n *= s; return n; } };

}  // namespace

struct fd_ndac {
  fd_ndac_config cfg;
  int hop = 1;
  std::vector<Param> params;
  std::map<std::string, std::vector<float>> host;
  std::map<std::string, float*> dev;
  std::vector<void*> allocs;
  // per-codebook device pointer tables for rvq_from_codes_kernel
  const float** d_wout = nullptr; const float** d_bout = nullptr; const float** d_cb = nullptr;
  std::vector<float*> cbn, c2;
  std::map<std::string, void*> packed;   // "<layer>.weight" -> split-bf16 A-operand copy for ndac_mfma.hip (the layers it supports)
  int precision = FD_NDAC_MFMA_DECODER;
  bool finalized = false;
};

namespace {

void build_params(fd_ndac* m) {
  const fd_ndac_config& c = m->cfg;
  auto& P = m->params;
  auto conv = [&](const std::string& n, int co, int ci, int k) { P.push_back({n + ".weight", {co, ci, k}}); P.push_back({n + ".bias", {co}}); };
  auto convT = [&](const std::string& n, int ci, int co, int k) { P.push_back({n + ".weight", {ci, co, k}}); P.push_back({n + ".bias", {co}}); };
  auto alpha = [&](const std::string& n, int ch) { P.push_back({n + ".alpha", {1, ch, 1}}); };
  auto res_unit = [&](const std::string& n, int dim) { alpha(n + ".block.0", dim); conv(n + ".block.1", dim, dim, 7); alpha(n + ".block.2", dim); conv(n + ".block.3", dim, dim, 1); };
  int d = c.encoder_dim;
  conv("encoder.block.0", d, 1, 7);
  for (int i = 0; i < c.n_encoder_rates; ++i) {
    d *= 2;
    const std::string p = "encoder.block." + std::to_string(i + 1);
    for (int j = 0; j < 3; ++j) res_unit(p + ".block." + std::to_string(j), d / 2);
    alpha(p + ".block.3", d / 2); conv(p + ".block.4", d, d / 2, 2 * c.encoder_rates[i]);
  }
  alpha("encoder.block." + std::to_string(c.n_encoder_rates + 1), d);
  conv("encoder.block." + std::to_string(c.n_encoder_rates + 2), c.latent_dim, d, 3);
  for (int i = 0; i < c.n_codebooks; ++i) {
    const std::string q = "quantizer.quantizers." + std::to_string(i);
    conv(q + ".in_proj", c.codebook_dim, c.latent_dim, 1); conv(q + ".out_proj", c.latent_dim, c.codebook_dim, 1);
    P.push_back({q + ".codebook.weight", {c.codebook_size, c.codebook_dim}});
  }
  const int D = c.decoder_dim;
  conv("decoder.model.0", D, c.latent_dim, 7);
  int od = D;
  for (int i = 0; i < c.n_decoder_rates; ++i) {
    const int idim = D >> i; od = D >> (i + 1);
    const std::string p = "decoder.model." + std::to_string(i + 1);
    alpha(p + ".block.0", idim); convT(p + ".block.1", idim, od, 2 * c.decoder_rates[i]);
    for (int j = 0; j < 3; ++j) res_unit(p + ".block." + std::to_string(j + 2), od);
  }
  alpha("decoder.model." + std::to_string(c.n_decoder_rates + 1), od);
  conv("decoder.model." + std::to_string(c.n_decoder_rates + 2), 1, od, 7);
}

// Inside the codec every Snake is applied ONCE, by the layer that produces the tensor (second output of its epilogue), never by the
// consumers: the activated input of a 768-channel convolution would otherwise be recomputed by each of its 24 channel-tile workgroups
// (sinf + a division per element).  Same function on the same float32 values: bit-identical to activating at the consumer.
struct Run {
  fd_ndac* m; hipStream_t st; int B; bool mfma;
  const float* P(const std::string& n) const { return m->dev.at(n); }
  const void* packed(const std::string& n) const {
    if (!mfma) return nullptr;
    auto it = m->packed.find(n + ".weight");
    return it == m->packed.end() ? nullptr : it->second;
  }
  // conv over an ALREADY ACTIVATED (or raw, for the first layer of a stack) input; raw result -> out (may be null), snake(result,
  // alpha_next) -> out_act (may be null)
  int conv(const float* x, const std::string& n, const float* res, float* out, float* out_act, const float* alpha_next, int Ci, int T, int Co, int K,
           int stride, int pad, int dil, int tanh_out = 0) const {
    if (const void* wp = packed(n); wp && !tanh_out)
      return fd_ndac_mfma_conv(x, wp, P(n + ".bias"), res, out, out_act, alpha_next, B, Ci, T, Co, K, stride, pad, dil, 0, st);
    return launch_conv1d(x, P(n + ".weight"), P(n + ".bias"), nullptr, res, out, B, Ci, T, Co, K, stride, pad, dil, tanh_out, st, /*wt*/ 1, out_act, alpha_next);
  }
  // ResidualUnit on (x_raw, x_act = snake(x, block.0.alpha)): y = x + conv1(snake(conv7_dil(x_act))); -> y_raw (may be null), snake(y, alpha_next)
  int res_unit(const float* x_raw, const float* x_act, const std::string& n, int dim, int T, int dil, float* tmp_act, float* y_raw, float* y_act,
               const float* alpha_next) const {
    FD_TRY(conv(x_act, n + ".block.1", nullptr, nullptr, tmp_act, P(n + ".block.2.alpha"), dim, T, dim, 7, 1, 3 * dil, dil));
    return conv(tmp_act, n + ".block.3", x_raw, y_raw, y_act, alpha_next, dim, T, dim, 1, 1, 0, 1);
  }
};

int ceil_half(int s) { return (s + 1) / 2; }

// widest intermediate (elements per batch item) of the encoder for input length L / of the decoder for T latent frames
long long enc_max_elems(const fd_ndac_config& c, long long L) {
  long long mx = 0, T = L; int d = c.encoder_dim;
  mx = (long long)d * T;
  for (int i = 0; i < c.n_encoder_rates; ++i) {
    d *= 2;
    const int s = c.encoder_rates[i];
    T = (T + 2 * ceil_half(s) - 2 * s) / s + 1;
    if ((long long)d * T > mx) mx = (long long)d * T;
  }
  return mx;
}
long long dec_out_len(const fd_ndac_config& c, long long T) {
  for (int i = 0; i < c.n_decoder_rates; ++i) { const int s = c.decoder_rates[i]; T = (T - 1) * s - 2 * ceil_half(s) + 2 * s; }
  return T;
}
long long dec_max_elems(const fd_ndac_config& c, long long T) {
  long long mx = (long long)c.decoder_dim * T;
  for (int i = 0; i < c.n_decoder_rates; ++i) {
    const int s = c.decoder_rates[i];
    T = (T - 1) * s - 2 * ceil_half(s) + 2 * s;
    const long long e = (long long)(c.decoder_dim >> (i + 1)) * T;
    if (e > mx) mx = e;
  }
  return mx;
}

int check_ready(const fd_ndac* m, const char* who) {
  FD_REQUIRE(m, "%s: null codec", who);
  if (!m->finalized) return fd_set_error(FD_ESTATE, "%s: codec not finalised (call fd_ndac_finalize)", who);
  return FD_OK;
}

int rvq_run(fd_ndac* m, float* residual /* in: z, destroyed */, int B, int T, int nq, float* zq, int* codes, float* latents, hipStream_t st) {
  const fd_ndac_config& c = m->cfg;
  FD_HIP(hipMemsetAsync(zq, 0, sizeof(float) * (size_t)B * c.latent_dim * T, st));
  for (int q = 0; q < nq; ++q) {
    const std::string n = "quantizer.quantizers." + std::to_string(q);
    RvqArgs a{residual, zq, codes, latents, m->dev.at(n + ".in_proj.weight"), m->dev.at(n + ".in_proj.bias"), m->dev.at(n + ".out_proj.weight"),
              m->dev.at(n + ".out_proj.bias"), m->dev.at(n + ".codebook.weight"), m->cbn[q], m->c2[q], B, c.latent_dim, T, c.codebook_size, c.codebook_dim, q, nq};
    hipLaunchKernelGGL(rvq_step_kernel, dim3(fd_cdiv(T, RT), B), dim3(256), 0, st, a);
  }
  FD_LAUNCH_CHECK();
  return FD_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int fd_conv1d(const float* x, const float* w, const float* bias, const float* alpha_in, const float* residual, float* out, int B, int Ci, int T,
                         int Co, int K, int stride, int padding, int dilation, int tanh_out, void* stream) {
  FD_REQUIRE(x && w && out, "fd_conv1d: null pointer");
  return launch_conv1d(x, w, bias, alpha_in, residual, out, B, Ci, T, Co, K, stride, padding, dilation, tanh_out, fd_stream(stream));
}

extern "C" int fd_conv_transpose1d(const float* x, const float* w, const float* bias, const float* alpha_in, float* out, int B, int Ci, int T, int Co, int K,
                                   int stride, int padding, void* stream) {
  FD_REQUIRE(x && w && out, "fd_conv_transpose1d: null pointer");
  return launch_convtr1d(x, w, bias, alpha_in, out, B, Ci, T, Co, K, stride, padding, fd_stream(stream));
}

extern "C" int fd_ndac_create(const fd_ndac_config* cfg, fd_ndac** out) {
  FD_REQUIRE(cfg && out, "fd_ndac_create: null pointer");
  FD_REQUIRE(cfg->encoder_dim > 0 && cfg->decoder_dim > 0 && cfg->latent_dim > 0, "fd_ndac_create: bad widths");
  FD_REQUIRE(cfg->n_encoder_rates >= 1 && cfg->n_encoder_rates <= 8 && cfg->n_decoder_rates >= 1 && cfg->n_decoder_rates <= 8, "fd_ndac_create: 1..8 rates per stack");
  FD_REQUIRE(cfg->decoder_dim % (1 << cfg->n_decoder_rates) == 0, "fd_ndac_create: decoder_dim must be divisible by 2^len(decoder_rates)");
  FD_REQUIRE(cfg->n_codebooks >= 1 && cfg->n_codebooks <= 64 && cfg->codebook_size >= 8 && cfg->codebook_dim >= 1 && cfg->codebook_dim <= 8,
             "fd_ndac_create: 1..64 codebooks of >= 8 entries and 1..8 dimensions (DAC: 8)");
  int hop = 1;
  for (int i = 0; i < cfg->n_encoder_rates; ++i) {
    FD_REQUIRE(cfg->encoder_rates[i] >= 1 && cfg->encoder_rates[i] <= 16, "fd_ndac_create: encoder rate %d out of range [1, 16]", cfg->encoder_rates[i]);
    hop *= cfg->encoder_rates[i];
  }
  for (int i = 0; i < cfg->n_decoder_rates; ++i)
    FD_REQUIRE(cfg->decoder_rates[i] >= 1 && cfg->decoder_rates[i] <= 16, "fd_ndac_create: decoder rate %d out of range [1, 16]", cfg->decoder_rates[i]);
  fd_ndac* m = new fd_ndac();
  m->cfg = *cfg; m->hop = hop;
  build_params(m);
  *out = m;
  return FD_OK;
}

extern "C" void fd_ndac_destroy(fd_ndac* m) {
  if (!m) return;
  for (void* p : m->allocs) (void)hipFree(p);
  delete m;
}

extern "C" int fd_ndac_hop_length(const fd_ndac* m) { return m ? m->hop : 0; }
extern "C" int fd_ndac_num_params(const fd_ndac* m) { return m ? (int)m->params.size() : 0; }

extern "C" int fd_ndac_param_info(const fd_ndac* m, int i, const char** name, int* ndim, int shape[3]) {
  FD_REQUIRE(m && name && ndim && shape && i >= 0 && i < (int)m->params.size(), "fd_ndac_param_info: bad arguments");
  const Param& p = m->params[i];
  *name = p.name.c_str(); *ndim = (int)p.shape.size();
  for (int k = 0; k < 3; ++k) shape[k] = k < (int)p.shape.size() ? p.shape[k] : 1;
  return FD_OK;
}

extern "C" int fd_ndac_set_param(fd_ndac* m, const char* name, const float* host_data, long long numel) {
  FD_REQUIRE(m && name && host_data, "fd_ndac_set_param: null pointer");
  FD_REQUIRE(!m->finalized, "fd_ndac_set_param: codec already finalised");
  for (const Param& p : m->params)
    if (p.name == name) {
      FD_REQUIRE(numel == p.numel(), "fd_ndac_set_param: '%s' expects %lld values (got %lld)", name, p.numel(), numel);
      m->host[name].assign(host_data, host_data + numel);
      return FD_OK;
    }
  return fd_set_error(FD_EINVAL, "fd_ndac_set_param: unknown parameter '%s'", name);
}

extern "C" int fd_ndac_finalize(fd_ndac* m, void* stream) {
  FD_REQUIRE(m, "fd_ndac_finalize: null codec");
  if (m->finalized) return FD_OK;
  hipStream_t st = fd_stream(stream);
  auto upload = [&](const std::vector<float>& h, float** out) -> int {
    float* d = nullptr;
    FD_HIP(hipMalloc(&d, sizeof(float) * h.size()));
    m->allocs.push_back(d);
    FD_HIP(hipMemcpyAsync(d, h.data(), sizeof(float) * h.size(), hipMemcpyHostToDevice, st));
    *out = d;
    return FD_OK;
  };
  for (const Param& p : m->params) {
    auto it = m->host.find(p.name);
    if (it == m->host.end()) return fd_set_error(FD_ESTATE, "fd_ndac_finalize: parameter '%s' missing", p.name.c_str());
    const bool enc_dec = p.name.rfind("encoder.", 0) == 0 || p.name.rfind("decoder.", 0) == 0;
    if (enc_dec && p.shape.size() == 3 && p.name.size() > 7 && p.name.compare(p.name.size() - 7, 7, ".weight") == 0) {
      // the conv stacks keep their weights as [Ci][K][Co] (output channel fastest: the kernels stage 32 consecutive channels per row);
      // state_dict layout is [Co][Ci][K] for Conv1d and [Ci][Co][K] for ConvTranspose1d (the only transposed ones: `.block.1` of a decoder block)
      size_t nblk = 0;
      for (size_t pos = p.name.find(".block."); pos != std::string::npos; pos = p.name.find(".block.", pos + 1)) ++nblk;
      const bool tr = p.name.rfind("decoder.model.", 0) == 0 && nblk == 1 && p.name.size() > 15 &&
                      p.name.compare(p.name.size() - 15, 15, ".block.1.weight") == 0;   // decoder.model.<i>.block.1 = WNConvTranspose1d
      const int d0 = p.shape[0], d1 = p.shape[1], K = p.shape[2];
      const int Ci = tr ? d0 : d1, Co = tr ? d1 : d0;
      std::vector<float> t((size_t)Ci * K * Co);
      for (int ci = 0; ci < Ci; ++ci)
        for (int k = 0; k < K; ++k)
          for (int co = 0; co < Co; ++co)
            t[((size_t)ci * K + k) * Co + co] = tr ? it->second[((size_t)ci * Co + co) * K + k] : it->second[((size_t)co * Ci + ci) * K + k];
      it->second.swap(t);
      const bool down = p.name.rfind("encoder.", 0) == 0 && nblk == 2 && p.name.size() > 15 &&
                        p.name.compare(p.name.size() - 15, 15, ".block.4.weight") == 0;   // encoder.block.<i>.block.4 = the strided WNConv1d (K = 2 s)
      const int stride = tr || down ? K / 2 : 1;
      if (fd_ndac_mfma_supported(Ci, Co, K, stride, tr || down || K == 1 ? 1 : 9, tr)) {   // (dilation 9: the widest staged tile of a K = 7 layer)
        const size_t nb = fd_ndac_mfma_packed_bytes(Ci, Co, K, stride, tr);
        std::vector<unsigned char> hp(nb);
        fd_ndac_mfma_pack(it->second.data(), Ci, Co, K, stride, tr, hp.data());
        void* dp = nullptr;
        FD_HIP(hipMalloc(&dp, nb));
        m->allocs.push_back(dp);
        FD_HIP(hipMemcpy(dp, hp.data(), nb, hipMemcpyHostToDevice));
        m->packed[p.name] = dp;
      }
    }
    float* d = nullptr;
    FD_TRY(upload(it->second, &d));
    m->dev[p.name] = d;
  }
  // L2-normalised codebooks and their squared norms, in the float32 operation order of the oracle (plain C float arithmetic on the
  // host: products and sums rounded separately -- this file is compiled without fast-math / contraction on the host side)
  const fd_ndac_config& c = m->cfg;
  std::vector<const float*> hw, hb, hc;
  for (int q = 0; q < c.n_codebooks; ++q) {
    const std::string n = "quantizer.quantizers." + std::to_string(q);
    const std::vector<float>& cb = m->host.at(n + ".codebook.weight");
    std::vector<float> cn(cb.size()), c2(c.codebook_size);
    for (int j = 0; j < c.codebook_size; ++j) {
      volatile float s = 0.f;
      for (int d = 0; d < c.codebook_dim; ++d) { volatile float pr = cb[(size_t)j * c.codebook_dim + d] * cb[(size_t)j * c.codebook_dim + d]; s = s + pr; }
      float nrm = sqrtf(s);
      if (nrm < 1e-12f) nrm = 1e-12f;
      volatile float s2 = 0.f;
      for (int d = 0; d < c.codebook_dim; ++d) {
        const float v = cb[(size_t)j * c.codebook_dim + d] / nrm;
        cn[(size_t)j * c.codebook_dim + d] = v;
        volatile float pr = v * v; s2 = s2 + pr;
      }
      c2[j] = s2;
    }
    float *dcn = nullptr, *dc2 = nullptr;
    FD_TRY(upload(cn, &dcn)); FD_TRY(upload(c2, &dc2));
    FD_HIP(hipStreamSynchronize(st));   // cn / c2 are locals
    m->cbn.push_back(dcn); m->c2.push_back(dc2);
    hw.push_back(m->dev.at(n + ".out_proj.weight")); hb.push_back(m->dev.at(n + ".out_proj.bias")); hc.push_back(m->dev.at(n + ".codebook.weight"));
  }
  auto upload_ptrs = [&](const std::vector<const float*>& h, const float*** out) -> int {
    const float** d = nullptr;
    FD_HIP(hipMalloc(&d, sizeof(float*) * h.size()));
    m->allocs.push_back((void*)d);
    FD_HIP(hipMemcpy((void*)d, h.data(), sizeof(float*) * h.size(), hipMemcpyHostToDevice));
    *out = d;
    return FD_OK;
  };
  FD_TRY(upload_ptrs(hw, &m->d_wout)); FD_TRY(upload_ptrs(hb, &m->d_bout)); FD_TRY(upload_ptrs(hc, &m->d_cb));
  FD_HIP(hipStreamSynchronize(st));
  m->host.clear();
  m->finalized = true;
  return FD_OK;
}

extern "C" int fd_ndac_set_precision(fd_ndac* m, int flags) {
  FD_REQUIRE(m, "fd_ndac_set_precision: null codec");
  FD_REQUIRE((flags & ~(FD_NDAC_MFMA_DECODER | FD_NDAC_MFMA_ENCODER)) == 0, "fd_ndac_set_precision: unknown flags 0x%x", flags);
  m->precision = flags;
  return FD_OK;
}
extern "C" int fd_ndac_get_precision(const fd_ndac* m) { return m ? m->precision : -1; }

extern "C" int fd_ndac_latent_frames(const fd_ndac* m, int L) {   // frames the encoder produces for L samples (L % hop == 0 -> L / hop)
  if (!m || L <= 0) return 0;
  long long T = L;
  for (int i = 0; i < m->cfg.n_encoder_rates; ++i) { const int s = m->cfg.encoder_rates[i]; T = (T + 2 * ceil_half(s) - 2 * s) / s + 1; }
  return (int)T;
}
extern "C" int fd_ndac_decoded_length(const fd_ndac* m, int T) { return m && T > 0 ? (int)dec_out_len(m->cfg, T) : 0; }

extern "C" size_t fd_ndac_workspace_bytes(const fd_ndac* m, int B, int L) {
  if (!m || B <= 0 || L <= 0) return 0;
  const long long T = fd_ndac_latent_frames(m, L);
  long long e = enc_max_elems(m->cfg, L), d = dec_max_elems(m->cfg, T > 0 ? T : 1);
  if (d > e) e = d;
  const long long lat = (long long)m->cfg.latent_dim * (T > 0 ? T : 1);
  if (lat > e) e = lat;
  return 5 * fd_align(sizeof(float) * (size_t)B * e) + 256;   // {raw, activated} of the current tensor, {raw, activated} of the next, one mid-unit tensor
}

extern "C" int fd_rvq_encode(fd_ndac* m, const float* z, int B, int T, int n_quantizers, float* z_q, int* codes, float* latents, void* ws, size_t ws_bytes,
                             void* stream) {
  FD_TRY(check_ready(m, "fd_rvq_encode"));
  FD_REQUIRE(z && z_q && codes && ws && B > 0 && T > 0, "fd_rvq_encode: bad arguments");
  const int nq = n_quantizers <= 0 || n_quantizers > m->cfg.n_codebooks ? m->cfg.n_codebooks : n_quantizers;
  const size_t need = sizeof(float) * (size_t)B * m->cfg.latent_dim * T;
  if (ws_bytes < need) return fd_set_error(FD_ENOMEM, "fd_rvq_encode: workspace %zu < required %zu bytes", ws_bytes, need);
  hipStream_t st = fd_stream(stream);
  FD_HIP(hipMemcpyAsync(ws, z, need, hipMemcpyDeviceToDevice, st));   // the residual chain works in place on a copy
  return rvq_run(m, (float*)ws, B, T, nq, z_q, codes, latents, st);
}

extern "C" int fd_rvq_from_codes(fd_ndac* m, const int* codes, int B, int n_quantizers, int T, float* z_q, void* stream) {
  FD_TRY(check_ready(m, "fd_rvq_from_codes"));
  FD_REQUIRE(codes && z_q && B > 0 && T > 0 && n_quantizers >= 1 && n_quantizers <= m->cfg.n_codebooks, "fd_rvq_from_codes: bad arguments");
  FromCodesArgs a{codes, z_q, m->d_wout, m->d_bout, m->d_cb, B, m->cfg.latent_dim, T, m->cfg.codebook_size, m->cfg.codebook_dim, n_quantizers};
  const long long n = (long long)B * a.D * T;
  hipLaunchKernelGGL(rvq_from_codes_kernel, dim3((unsigned)(n / 256 + 1 > 65535 ? 65535 : n / 256 + 1)), dim3(256), 0, fd_stream(stream), a);
  FD_LAUNCH_CHECK();
  return FD_OK;
}

extern "C" int fd_ndac_encode(fd_ndac* m, const float* x, int B, int L, int n_quantizers, float* z_q, int* codes, float* latents, void* ws,
                              size_t ws_bytes, void* stream) {
  FD_TRY(check_ready(m, "fd_ndac_encode"));
  FD_REQUIRE(x && z_q && codes && ws && B > 0 && L > 0, "fd_ndac_encode: bad arguments");
  FD_REQUIRE(L % m->hop == 0, "fd_ndac_encode: pad the waveform to a multiple of the hop length %d first (DAC.preprocess); got %d samples", m->hop, L);
  const size_t need = fd_ndac_workspace_bytes(m, B, L);
  if (ws_bytes < need) return fd_set_error(FD_ENOMEM, "fd_ndac_encode: workspace %zu < required %zu bytes", ws_bytes, need);
  const fd_ndac_config& c = m->cfg;
  hipStream_t st = fd_stream(stream);
  const size_t slot = (need - 256) / 5;
  float* buf[5];
  for (int i = 0; i < 5; ++i) buf[i] = (float*)((char*)ws + i * slot);
  Run r{m, st, B, (m->precision & FD_NDAC_MFMA_ENCODER) != 0};
  int T = L, d = c.encoder_dim;
  float *cur_raw = buf[0], *cur_act = buf[1], *nxt_raw = buf[2], *nxt_act = buf[3], *tmp = buf[4];
  auto alpha_of = [&](const std::string& n) { return r.P(n + ".alpha"); };
  // block.0: WNConv1d(1, d, 7) on the raw audio; its output enters EncoderBlock 1 = ResidualUnit 0 (needs raw + snake(block.0.alpha))
  FD_TRY(r.conv(x, "encoder.block.0", nullptr, cur_raw, cur_act, alpha_of("encoder.block.1.block.0.block.0"), 1, T, d, 7, 1, 3, 1));
  for (int i = 0; i < c.n_encoder_rates; ++i) {
    const std::string p = "encoder.block." + std::to_string(i + 1);
    const int dil[3] = {1, 3, 9};
    for (int j = 0; j < 3; ++j) {   // unit j -> (raw, activated for unit j + 1 / for the block's Snake in front of the strided conv)
      const float* an = j < 2 ? alpha_of(p + ".block." + std::to_string(j + 1) + ".block.0") : alpha_of(p + ".block.3");
      FD_TRY(r.res_unit(cur_raw, cur_act, p + ".block." + std::to_string(j), d, T, dil[j], tmp, j < 2 ? nxt_raw : nullptr, nxt_act, an));
      std::swap(cur_raw, nxt_raw); std::swap(cur_act, nxt_act);
    }
    const int s = c.encoder_rates[i];
    const bool last = i + 1 == c.n_encoder_rates;
    const float* an = last ? alpha_of("encoder.block." + std::to_string(c.n_encoder_rates + 1))
                           : alpha_of("encoder.block." + std::to_string(i + 2) + ".block.0.block.0");
    FD_TRY(r.conv(cur_act, p + ".block.4", nullptr, last ? nullptr : nxt_raw, nxt_act, an, d, T, 2 * d, 2 * s, s, ceil_half(s), 1));
    std::swap(cur_raw, nxt_raw); std::swap(cur_act, nxt_act);
    T = (T + 2 * ceil_half(s) - 2 * s) / s + 1;
    d *= 2;
  }
  const std::string last = "encoder.block." + std::to_string(c.n_encoder_rates + 2);
  FD_TRY(r.conv(cur_act, last, nullptr, nxt_raw, nullptr, nullptr, d, T, c.latent_dim, 3, 1, 1, 1));
  float* const zbuf = nxt_raw;
  const int nq = n_quantizers <= 0 || n_quantizers > c.n_codebooks ? c.n_codebooks : n_quantizers;
  return rvq_run(m, zbuf, B, T, nq, z_q, codes, latents, st);
}

extern "C" int fd_ndac_decode(fd_ndac* m, const float* z, int B, int T, float* audio, void* ws, size_t ws_bytes, void* stream) {
  FD_TRY(check_ready(m, "fd_ndac_decode"));
  FD_REQUIRE(z && audio && ws && B > 0 && T > 0, "fd_ndac_decode: bad arguments");
  const fd_ndac_config& c = m->cfg;
  const size_t slot = fd_align(sizeof(float) * (size_t)B * dec_max_elems(c, T));
  if (ws_bytes < 5 * slot) return fd_set_error(FD_ENOMEM, "fd_ndac_decode: workspace %zu < required %zu bytes", ws_bytes, 5 * slot);
  hipStream_t st = fd_stream(stream);
  float* buf[5];
  for (int i = 0; i < 5; ++i) buf[i] = (float*)((char*)ws + i * slot);
  Run r{m, st, B, (m->precision & FD_NDAC_MFMA_DECODER) != 0};
  float *cur_raw = buf[0], *cur_act = buf[1], *nxt_raw = buf[2], *nxt_act = buf[3], *tmp = buf[4];
  auto alpha_of = [&](const std::string& n) { return r.P(n + ".alpha"); };
  // model.0: WNConv1d(latent, D, 7); DecoderBlock 1 starts with a Snake: only the activated output is needed
  FD_TRY(r.conv(z, "decoder.model.0", nullptr, nullptr, cur_act, alpha_of("decoder.model.1.block.0"), c.latent_dim, T, c.decoder_dim, 7, 1, 3, 1));
  int od = c.decoder_dim;
  for (int i = 0; i < c.n_decoder_rates; ++i) {
    const int idim = c.decoder_dim >> i, s = c.decoder_rates[i];
    od = c.decoder_dim >> (i + 1);
    const std::string p = "decoder.model." + std::to_string(i + 1);
    if (const void* wp = r.packed(p + ".block.1"))
      FD_TRY(fd_ndac_mfma_conv(cur_act, wp, r.P(p + ".block.1.bias"), nullptr, nxt_raw, nxt_act, alpha_of(p + ".block.2.block.0"), B, idim, T, od, 2 * s, s,
                               ceil_half(s), 1, 1, st));
    else
      FD_TRY(launch_convtr1d(cur_act, r.P(p + ".block.1.weight"), r.P(p + ".block.1.bias"), nullptr, nxt_raw, B, idim, T, od, 2 * s, s, ceil_half(s), st, /*wt*/ 1,
                             nxt_act, alpha_of(p + ".block.2.block.0")));
    std::swap(cur_raw, nxt_raw); std::swap(cur_act, nxt_act);
    T = (T - 1) * s - 2 * ceil_half(s) + 2 * s;
    const int dil[3] = {1, 3, 9};
    for (int j = 0; j < 3; ++j) {
      const bool last_unit = j == 2;
      const float* an = !last_unit ? alpha_of(p + ".block." + std::to_string(j + 3) + ".block.0")
                                   : (i + 1 < c.n_decoder_rates ? alpha_of("decoder.model." + std::to_string(i + 2) + ".block.0")
                                                                : alpha_of("decoder.model." + std::to_string(c.n_decoder_rates + 1)));
      FD_TRY(r.res_unit(cur_raw, cur_act, p + ".block." + std::to_string(j + 2), od, T, dil[j], tmp, last_unit ? nullptr : nxt_raw, nxt_act, an));
      std::swap(cur_raw, nxt_raw); std::swap(cur_act, nxt_act);
    }
  }
  const std::string last = "decoder.model." + std::to_string(c.n_decoder_rates + 2);
  return r.conv(cur_act, last, nullptr, audio, nullptr, nullptr, od, T, 1, 7, 1, 3, 1, /*tanh*/ 1);
}
