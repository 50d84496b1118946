// stft.hip -- front/back end of FlowModel.enhance on the GPU.
//
// Forward (model.py:129-163): normalize_noisy (util/other.py:55-82) -> torch.stft(n_fft=1534, hop=384,
// symmetric Hann, center/reflect, onesided) (feature_extractors.py:86-96) -> amplitude compression
// (:118-128) -> zero pad frames to a multiple of 64 (util/other.py:25-52).
// Inverse (model.py:165-190): slice -> decompression (:130-139) -> torch.istft(length) (:98-109) -> * normfac.
//
// n_fft = 1534 = 2*13*59 is FFT-hostile, and the whole transform is ~10 GFLOP per call (0.01 % of the
// network), so the DFT is evaluated as an exact-f32 GEMM on the matrix cores (v_mfma_f32_32x32x2_f32)
// against a precomputed [1536 x 1536] (window-folded) DFT matrix that stays L2/MALL resident:
//   frames[B*T][1536] x Dt[1536][2*768]  ->  spec[B*T][re,im interleaved]
//   spec'[B*T][1536]  x E[1536][1536]    ->  windowed time frames, then a gather overlap-add.
#include <math.h>

#include <vector>

#include "common.h"
#include "internal.h"

struct fd_stft_plan {
  int n_fft, hop, n_freq, kpad;
  float* Dt = nullptr;   // [kpad][kpad]   forward: rows = sample k (window folded), cols = 2f (re), 2f+1 (im)
  float* E = nullptr;    // [kpad][kpad]   inverse: rows = 2f / 2f+1, cols = sample n (window and 1/N folded)
  float* w2 = nullptr;   // [n_fft]        window^2 (overlap-add envelope)
  // optional per-kernel timing (fd_stft_plan_profile): events 0..3 bracket {absmax + framing | DFT GEMM | compression} of the LAST
  // forward call, 4..7 {decompression | inverse DFT GEMM | overlap-add} of the last inverse call
  bool prof = false;
  hipEvent_t ev[8] = {};
  int calls[2] = {0, 0};
};

namespace {

// Ragged batches (fd_*_ragged): `lens` (device int32 [B], may be null) holds every clip's own sample count; clip b then occupies
// the first lens[b] samples of its row of L (= the longest clip's length) and has its own frame count 1 + lens[b] / hop.  Every
// kernel below does for clip b exactly the arithmetic it would do in a call with that clip alone (same reflect padding, same
// frames, same overlap-add envelope), so a clip's output does not depend on its neighbours.  The value is clamped into what the
// buffers hold: a wrong length can give a wrong waveform, never an out-of-bounds access.
__device__ __forceinline__ int clip_len(const int* __restrict__ lens, int b, int L, int n_fft) {
  if (!lens) return L;
  const int l = lens[b], lo = n_fft / 2 + 1;
  return l < lo ? lo : (l > L ? L : l);
}

// per-clip max |y| -> normfac (isclose(normfac, 0) -> 1), one block per clip
__global__ __launch_bounds__(1024) void absmax_kernel(const float* __restrict__ y, const int* __restrict__ lens, int L, int n_fft, int normalize,
                                                      float* __restrict__ normfac) {
  const int b = blockIdx.x;
  const int Lb = clip_len(lens, b, L, n_fft);
  float m = 0.f;
  if (normalize)
    for (int i = threadIdx.x; i < Lb; i += 1024) m = fmaxf(m, fabsf(y[(size_t)b * L + i]));
  m = fd_wave_max(m);
  __shared__ float red[16];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    float r = 0.f;
    for (int i = 0; i < 16; ++i) r = fmaxf(r, red[i]);
    if (!normalize) r = 1.f;
    normfac[b] = (fabsf(r) <= 1e-8f) ? 1.0f : r;  // torch.isclose(normfac, 0) with default atol
  }
}

// frames[b*T + t][k] = y[b][reflect(hop*t + k - n_fft/2)] / normfac[b]; columns >= n_fft are zero (ragged: also the frames a
// shorter clip does not have)
__global__ void frame_kernel(const float* __restrict__ y, const int* __restrict__ lens, const float* __restrict__ normfac, float* __restrict__ frames, int B,
                             int L, int T, int n_fft, int hop, int kpad) {
  const long long total = (long long)B * T * kpad;
  const int pad = n_fft / 2;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i % kpad);
    const long long fr = i / kpad;
    const int t = (int)(fr % T), b = (int)(fr / T);
    const int Lb = clip_len(lens, b, L, n_fft);
    float v = 0.f;
    if (k < n_fft && t < 1 + Lb / hop) {
      int s = hop * t + k - pad;
      s = s < 0 ? -s : s;
      s = s >= Lb ? 2 * (Lb - 1) - s : s;
      v = y[(size_t)b * L + s] / normfac[b];
    }
    frames[i] = v;
  }
}

// C[M][N] = A[M][K] * B[K][N], row-major f32, N and K multiples of 128 / 16.  128 x BN x 16 tiles of v_mfma_f32_32x32x2_f32 (exact f32
// products, f32 accumulation, k ascending: the result does not depend on BN).  BN = 128: 4 waves as 2 x 2, each 2 x 2 MFMA tiles;
// BN = 32: 4 waves as 4 x 1, one tile each -- 4 x the workgroups for the small grids (one clip: 12 workgroups of 128 x 128 took
// 154 us, MFMA-bound per workgroup at 32 f32 MFMAs per k-block).
template <int BN>
__global__ __launch_bounds__(256) void sgemm_mfma_kernel(const float* __restrict__ A, const float* __restrict__ Bm, float* __restrict__ C,
                                                         int M, int N, int K) {
  constexpr int BM = 128, BK = 16, AP = BK + 1, BP = BN + 4;
  constexpr int WN = BN == 128 ? 2 : 1, WM = 4 / WN, TI = BM / (32 * WM), TJ = BN / (32 * WN);   // wave grid, MFMA tiles per wave
  __shared__ float As[BM * AP];
  __shared__ float Bs[BK * BP];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int ar = t >> 1, ac = (t & 1) * 8;     // A tile: 128 rows x 16 cols, 8 floats per thread
  constexpr int BV = BN * BK / 256;             // B tile: 16 rows x BN cols, 8 (BN = 128) or 2 (BN = 32) floats per thread
  const int br = t / (BN / BV), bc = (t % (BN / BV)) * BV;
  const bool arow_ok = (m0 + ar) < M;
  // the global loads of k-block i + 1 are issued before the MFMAs of block i (register double buffer)
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
  float bv[BV];
  auto load_block = [&](int k0) {
    if (arow_ok) {
      const float* ap = A + (size_t)(m0 + ar) * K + k0 + ac;
      a0 = *reinterpret_cast<const f32x4*>(ap);
      a1 = *reinterpret_cast<const f32x4*>(ap + 4);
    }
    const float* bp = Bm + (size_t)(k0 + br) * N + n0 + bc;
    if constexpr (BV == 8) {
      const f32x4 q0 = *reinterpret_cast<const f32x4*>(bp), q1 = *reinterpret_cast<const f32x4*>(bp + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { bv[e] = q0[e]; bv[4 + e] = q1[e]; }
    } else {
      const float2 q = *reinterpret_cast<const float2*>(bp);
      bv[0] = q.x; bv[1] = q.y;
    }
  };
  load_block(0);
  for (int k0 = 0; k0 < K; k0 += BK) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) { As[ar * AP + ac + e] = a0[e]; As[ar * AP + ac + 4 + e] = a1[e]; }
#pragma unroll
    for (int e = 0; e < BV; ++e) Bs[br * BP + bc + e] = bv[e];
    __syncthreads();
    load_block(k0 + BK < K ? k0 + BK : k0);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float af[TI], bf[TJ];
#pragma unroll
      for (int i = 0; i < TI; ++i) af[i] = As[(wm * 32 * TI + i * 32 + l31) * AP + kk + lh];
#pragma unroll
      for (int j = 0; j < TJ; ++j) bf[j] = Bs[(kk + lh) * BP + wn * 32 * TJ + j * 32 + l31];
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 * TI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int n = n0 + wn * 32 * TJ + j * 32 + l31;
        if (m < M) C[(size_t)m * N + n] = acc[i][j][r];
      }
}

// narrow tiles while the wide ones would leave compute units idle
static void launch_sgemm(const float* A, const float* Bm, float* C, int M, int N, int K, hipStream_t st) {
  const int rows = fd_cdiv(M, 128);
  if ((long long)rows * (N / 128) >= 512) hipLaunchKernelGGL(sgemm_mfma_kernel<128>, dim3(N / 128, rows), dim3(256), 0, st, A, Bm, C, M, N, K);
  else hipLaunchKernelGGL(sgemm_mfma_kernel<32>, dim3(N / 32, rows), dim3(256), 0, st, A, Bm, C, M, N, K);
}

// spec[b*T+t][2f,2f+1] -> Y[b][f][t] = beta * |X|^alpha * exp(j angle X); frames t >= T are zero padding
__global__ void compress_kernel(const float* __restrict__ spec, const int* __restrict__ lens, float2* __restrict__ Y, int B, int F, int T, int T_pad,
                                int kpad, int L, int n_fft, int hop, float alpha, float beta) {
  const long long total = (long long)B * F * T_pad;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int t = (int)(i % T_pad);
    const long long r = i / T_pad;
    const int f = (int)(r % F), b = (int)(r / F);
    float2 o = {0.f, 0.f};
    if (t < (lens ? 1 + clip_len(lens, b, L, n_fft) / hop : T)) {
      const float2 x = *reinterpret_cast<const float2*>(spec + ((size_t)b * T + t) * kpad + 2 * f);
      float re = x.x, im = x.y;
      if (alpha != 1.0f) {
        const float mag = powf(hypotf(re, im), alpha);
        const float th = atan2f(im, re);
        float sn, cs;
        sincosf(th, &sn, &cs);
        re = mag * cs; im = mag * sn;
      }
      o.x = re * beta; o.y = im * beta;
    }
    Y[i] = o;
  }
}

// X[b][f][t] (t < T) -> Z[b*T+t][2f,2f+1] = |X/beta|^(1/alpha) exp(j angle(X/beta))
__global__ void decompress_kernel(const float2* __restrict__ X, const int* __restrict__ lens, float* __restrict__ Z, int B, int F, int T, int T_pad,
                                  int kpad, int L, int n_fft, int hop, float alpha, float beta) {
  const long long total = (long long)B * T * (kpad / 2);
  const float inv_alpha = 1.0f / alpha;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int f = (int)(i % (kpad / 2));
    const long long fr = i / (kpad / 2);
    const int t = (int)(fr % T), b = (int)(fr / T);
    float2 o = {0.f, 0.f};
    if (f < F && (!lens || t < 1 + clip_len(lens, b, L, n_fft) / hop)) {
      const float2 x = X[((size_t)b * F + f) * T_pad + t];
      float re = x.x / beta, im = x.y / beta;
      if (alpha != 1.0f) {
        const float mag = powf(hypotf(re, im), inv_alpha);
        const float th = atan2f(im, re);
        float sn, cs;
        sincosf(th, &sn, &cs);
        re = mag * cs; im = mag * sn;
      }
      o.x = re; o.y = im;
    }
    *reinterpret_cast<float2*>(Z + (size_t)fr * kpad + 2 * f) = o;
  }
}

// y[b][s] = normfac[b] * (sum_t FR[b*T+t][s + n_fft/2 - hop*t]) / (sum_t w^2[...]),  s < L; beyond the
// synthesised length the output is zero (torch.istft pads with zeros)
__global__ void overlap_add_kernel(const float* __restrict__ FR, const int* __restrict__ lens, const float* __restrict__ w2,
                                   const float* __restrict__ normfac, float* __restrict__ y, int B, int T, int L, int n_fft, int hop, int kpad) {
  const long long total = (long long)B * L;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int s = (int)(i % L), b = (int)(i / L);
    // ragged: clip b was synthesised from ITS frame count (torch.istft(length = its own length)); the row's tail stays zero
    const int Lb = clip_len(lens, b, L, n_fft), Tb = lens ? 1 + Lb / hop : T;
    const int total_len = n_fft + hop * (Tb - 1);
    const int p = s + n_fft / 2;
    float o = 0.f;
    if (p < total_len && s < Lb) {
      int t_hi = p / hop; if (t_hi > Tb - 1) t_hi = Tb - 1;
      int t_lo = (p - (n_fft - 1) + hop - 1) / hop; if (t_lo < 0) t_lo = 0;
      float acc = 0.f, env = 0.f;
      for (int t = t_lo; t <= t_hi; ++t) {
        const int n = p - hop * t;
        acc += FR[((size_t)b * T + t) * kpad + n];
        env += w2[n];
      }
      o = acc / env;
      if (normfac) o *= normfac[b];
    }
    y[i] = o;
  }
}

inline int grid_cap(long long n) { long long g = (n + 255) / 256; return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g)); }

}  // namespace

extern "C" int fd_stft_plan_create(int n_fft, int hop, fd_stft_plan** out) {
  FD_REQUIRE(n_fft > 0 && n_fft % 2 == 0 && hop > 0, "stft plan: n_fft must be even and positive");
  fd_stft_plan* p = new fd_stft_plan();
  p->n_fft = n_fft; p->hop = hop; p->n_freq = n_fft / 2 + 1;
  p->kpad = (2 * p->n_freq + 127) / 128 * 128;  // 1536 for n_fft 1534
  if (p->kpad < n_fft) p->kpad = (n_fft + 127) / 128 * 128;
  const int K = p->kpad;
  std::vector<float> Dt((size_t)K * K, 0.f), E((size_t)K * K, 0.f), w2(n_fft);
  std::vector<double> w(n_fft);
  for (int k = 0; k < n_fft; ++k) {
    w[k] = 0.5 - 0.5 * cos(2.0 * M_PI * k / (n_fft - 1));  // torch.signal.windows.hann(sym=True), feature_extractors.py:73-75
    const float wf = (float)w[k];                           // the reference window is a float32 tensor
    w[k] = (double)wf;
    w2[k] = wf * wf;
  }
  for (int k = 0; k < n_fft; ++k)
    for (int f = 0; f < p->n_freq; ++f) {
      const long long ph = ((long long)k * f) % n_fft;  // exact phase reduction
      const double ang = 2.0 * M_PI * (double)ph / n_fft;
      Dt[(size_t)k * K + 2 * f] = (float)(w[k] * cos(ang));
      Dt[(size_t)k * K + 2 * f + 1] = (float)(-w[k] * sin(ang));
      const double cf = (f == 0 || f == n_fft / 2) ? 1.0 : 2.0;
      E[(size_t)(2 * f) * K + k] = (float)(w[k] * cf * cos(ang) / n_fft);
      E[(size_t)(2 * f + 1) * K + k] = (float)(-w[k] * cf * sin(ang) / n_fft);
    }
  FD_HIP(hipMalloc(&p->Dt, sizeof(float) * K * K));
  FD_HIP(hipMalloc(&p->E, sizeof(float) * K * K));
  FD_HIP(hipMalloc(&p->w2, sizeof(float) * n_fft));
  FD_HIP(hipMemcpy(p->Dt, Dt.data(), sizeof(float) * K * K, hipMemcpyHostToDevice));
  FD_HIP(hipMemcpy(p->E, E.data(), sizeof(float) * K * K, hipMemcpyHostToDevice));
  FD_HIP(hipMemcpy(p->w2, w2.data(), sizeof(float) * n_fft, hipMemcpyHostToDevice));
  *out = p;
  return FD_OK;
}

extern "C" void fd_stft_plan_destroy(fd_stft_plan* p) {
  if (!p) return;
  (void)hipFree(p->Dt); (void)hipFree(p->E); (void)hipFree(p->w2);
  for (hipEvent_t e : p->ev) if (e) (void)hipEventDestroy(e);
  delete p;
}

extern "C" int fd_stft_plan_profile(fd_stft_plan* p, int enable) {
  FD_REQUIRE(p, "fd_stft_plan_profile: null plan");
  if (enable)
    for (hipEvent_t& e : p->ev) if (!e) FD_HIP(hipEventCreate(&e));
  p->prof = enable != 0;
  p->calls[0] = p->calls[1] = 0;
  return FD_OK;
}

extern "C" int fd_stft_plan_profile_read(fd_stft_plan* p, double* ms6, int* calls2) {
  FD_REQUIRE(p && ms6, "fd_stft_plan_profile_read: null pointer");
  for (int i = 0; i < 6; ++i) ms6[i] = 0.0;
  for (int half = 0; half < 2; ++half) {
    if (!p->calls[half]) continue;
    FD_HIP(hipEventSynchronize(p->ev[4 * half + 3]));
    for (int i = 0; i < 3; ++i) {
      float e = 0.f;
      FD_HIP(hipEventElapsedTime(&e, p->ev[4 * half + i], p->ev[4 * half + i + 1]));
      ms6[3 * half + i] = e;
    }
  }
  if (calls2) { calls2[0] = p->calls[0]; calls2[1] = p->calls[1]; }
  return FD_OK;
}

size_t fd_stft_ws_bytes(int B, int L, int n_fft, int hop) {
  const int T = 1 + L / hop;
  const size_t kpad = ((size_t)(n_fft + 2) + 127) / 128 * 128;
  return 2 * fd_align(sizeof(float) * (size_t)B * T * kpad) + 256;
}

int fd_stft_forward(fd_stft_plan* p, const float* y, const int* lens, int B, int L, float alpha, float beta, int normalize, float* normfac, float* Y,
                    int T_pad, void* ws, size_t ws_bytes, hipStream_t st) {
  const int T = 1 + L / p->hop, K = p->kpad;
  FD_REQUIRE(L > p->n_fft / 2, "stft: clip of %d samples is too short for reflect padding of %d", L, p->n_fft / 2);
  FD_REQUIRE(T_pad >= T, "stft: T_pad %d < T %d", T_pad, T);
  if (ws_bytes < fd_stft_ws_bytes(B, L, p->n_fft, p->hop)) return fd_set_error(FD_ENOMEM, "stft: workspace too small");
  float* frames = reinterpret_cast<float*>(ws);
  float* spec = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + fd_align(sizeof(float) * (size_t)B * T * K));
  const int M = B * T;
  auto mark = [&](int i) { if (p->prof) (void)hipEventRecord(p->ev[i], st); };
  mark(0);
  hipLaunchKernelGGL(absmax_kernel, dim3(B), dim3(1024), 0, st, y, lens, L, p->n_fft, normalize, normfac);
  hipLaunchKernelGGL(frame_kernel, dim3(grid_cap((long long)M * K)), dim3(256), 0, st, y, lens, normfac, frames, B, L, T, p->n_fft, p->hop, K);
  mark(1);
  launch_sgemm(frames, p->Dt, spec, M, K, K, st);
  mark(2);
  hipLaunchKernelGGL(compress_kernel, dim3(grid_cap((long long)B * p->n_freq * T_pad)), dim3(256), 0, st, spec, lens, (float2*)Y, B, p->n_freq, T,
                     T_pad, K, L, p->n_fft, p->hop, alpha, beta);
  mark(3);
  if (p->prof) ++p->calls[0];
  FD_LAUNCH_CHECK();
  return FD_OK;
}

int fd_stft_inverse(fd_stft_plan* p, const float* X, const int* lens, int B, int T, int T_pad, float alpha, float beta, const float* normfac, float* y,
                    int L, void* ws, size_t ws_bytes, hipStream_t st) {
  const int K = p->kpad;
  FD_REQUIRE(T >= 1 && T_pad >= T, "istft: bad frame counts");
  const size_t need = 2 * fd_align(sizeof(float) * (size_t)B * T * K);
  if (ws_bytes < need) return fd_set_error(FD_ENOMEM, "istft: workspace too small (%zu < %zu)", ws_bytes, need);
  float* Z = reinterpret_cast<float*>(ws);
  float* FR = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + fd_align(sizeof(float) * (size_t)B * T * K));
  const int M = B * T;
  auto mark = [&](int i) { if (p->prof) (void)hipEventRecord(p->ev[i], st); };
  mark(4);
  hipLaunchKernelGGL(decompress_kernel, dim3(grid_cap((long long)M * (K / 2))), dim3(256), 0, st, (const float2*)X, lens, Z, B, p->n_freq, T, T_pad, K,
                     L, p->n_fft, p->hop, alpha, beta);
  mark(5);
  launch_sgemm(Z, p->E, FR, M, K, K, st);
  mark(6);
  hipLaunchKernelGGL(overlap_add_kernel, dim3(grid_cap((long long)B * L)), dim3(256), 0, st, FR, lens, p->w2, normfac, y, B, T, L, p->n_fft, p->hop, K);
  mark(7);
  if (p->prof) ++p->calls[1];
  FD_LAUNCH_CHECK();
  return FD_OK;
}

// stand-alone amplitude compression / its inverse on a complex tensor (CompressAmplitudesAndScale.forward / .invert,
// feature_extractors.py:118-139; on the hot path the same arithmetic is fused into compress_kernel / decompress_kernel)
__global__ void compress_spec_kernel(const float2* __restrict__ X, float2* __restrict__ Y, long long n, float alpha, float beta, int inverse) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float re = X[i].x, im = X[i].y;
    if (inverse) { re /= beta; im /= beta; }
    const float e = inverse ? 1.0f / alpha : alpha;
    if (e != 1.0f) {
      const float mag = powf(hypotf(re, im), e);
      const float th = atan2f(im, re);
      float sn, cs;
      sincosf(th, &sn, &cs);
      re = mag * cs; im = mag * sn;
    }
    if (!inverse) { re *= beta; im *= beta; }
    Y[i] = float2{re, im};
  }
}

extern "C" int fd_compress_spec(const float* X, float* Y, long long n, float alpha, float beta, int inverse, void* stream) {
  FD_REQUIRE(X && Y && n > 0 && alpha > 0.f && beta > 0.f, "fd_compress_spec: bad arguments");
  hipLaunchKernelGGL(compress_spec_kernel, dim3(grid_cap(n)), dim3(256), 0, fd_stream(stream), (const float2*)X, (float2*)Y, n, alpha, beta, inverse);
  FD_LAUNCH_CHECK();
  return FD_OK;
}

// ---- stand-alone C ABI: the caller owns the plan (no hidden state, nothing allocates on the hot path) -------------------
extern "C" size_t fd_stft_workspace_bytes(int B, int L, int n_fft, int hop) { return fd_stft_ws_bytes(B, L, n_fft, hop); }
extern "C" int fd_num_frames(int L, int hop) { return 1 + L / hop; }
extern "C" int fd_padded_frames(int T) { return (T % 64 == 0) ? T : T + (64 - T % 64); }

extern "C" int fd_stft_compress(const fd_stft_plan* plan, const float* y, int B, int L, float alpha, float beta, int normalize,
                                float* normfac, float* Y, int T_pad, void* ws, size_t ws_bytes, void* stream) {
  FD_REQUIRE(plan && y && normfac && Y && ws && B > 0 && L > 0, "fd_stft_compress: bad arguments");
  return fd_stft_forward(const_cast<fd_stft_plan*>(plan), y, nullptr, B, L, alpha, beta, normalize, normfac, Y, T_pad, ws, ws_bytes, fd_stream(stream));
}

// Ragged batch: clip b holds lengths[b] samples (device int32 [B]; n_fft/2 < lengths[b] <= L) in its row of L; its frames
// t >= 1 + lengths[b]/hop are zero, as pad_spec leaves them (util/other.py:25-52).  Bit-identical per clip to a call with that clip alone.
extern "C" int fd_stft_compress_ragged(const fd_stft_plan* plan, const float* y, const int* lengths, int B, int L, float alpha, float beta,
                                       int normalize, float* normfac, float* Y, int T_pad, void* ws, size_t ws_bytes, void* stream) {
  FD_REQUIRE(plan && y && lengths && normfac && Y && ws && B > 0 && L > 0, "fd_stft_compress_ragged: bad arguments");
  return fd_stft_forward(const_cast<fd_stft_plan*>(plan), y, lengths, B, L, alpha, beta, normalize, normfac, Y, T_pad, ws, ws_bytes, fd_stream(stream));
}

extern "C" int fd_decompress_istft(const fd_stft_plan* plan, const float* X, int B, int T, int T_pad, float alpha, float beta,
                                   const float* normfac, float* y, int L, void* ws, size_t ws_bytes, void* stream) {
  FD_REQUIRE(plan && X && y && ws && B > 0 && L > 0, "fd_decompress_istft: bad arguments");
  return fd_stft_inverse(const_cast<fd_stft_plan*>(plan), X, nullptr, B, T, T_pad, alpha, beta, normfac, y, L, ws, ws_bytes, fd_stream(stream));
}

// Ragged batch: T = frames of the LONGEST clip (1 + L/hop); clip b is synthesised from its own 1 + lengths[b]/hop frames with
// torch.istft(length = lengths[b]) semantics, samples [lengths[b], L) of its row are zero.
extern "C" int fd_decompress_istft_ragged(const fd_stft_plan* plan, const float* X, const int* lengths, int B, int T, int T_pad, float alpha,
                                          float beta, const float* normfac, float* y, int L, void* ws, size_t ws_bytes, void* stream) {
  FD_REQUIRE(plan && X && lengths && y && ws && B > 0 && L > 0, "fd_decompress_istft_ragged: bad arguments");
  FD_REQUIRE(T == 1 + L / plan->hop, "fd_decompress_istft_ragged: T must be the frame count of the row length L (1 + L / hop = %d, got %d)", 1 + L / plan->hop, T);
  return fd_stft_inverse(const_cast<fd_stft_plan*>(plan), X, lengths, B, T, T_pad, alpha, beta, normfac, y, L, ws, ws_bytes, fd_stream(stream));
}
