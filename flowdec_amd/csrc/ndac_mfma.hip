// ndac_mfma.hip -- the NDAC codec's wide 1-D convolutions on the gfx950 matrix cores (v_mfma_f32_32x32x16_bf16), at float32
// tolerance: both operands are split into two bf16 terms (x = xh + xl, w = wh + wl) and three products are accumulated in f32
// (wl xh + wh xl + wh xh; the dropped wl xl term is 2^-16 relative), the same scheme as the post-filter's `bf16x3` mode.
// The vector-ALU kernels of ndac.hip stay the EXACT path (bit-defined operation order: the encoder's code indices); this file is
// the decoder's default path (dac.DAC.decode is a tolerance comparison in any implementation: cuDNN does not define an order).
//
// One kernel, implicit GEMM  out[co][n] = sum_tap sum_ci W_tap[co][ci] x[ci][pos(n, tap)]:
//   Conv1d (stride 1, dilation d)        taps k = 0..K-1, pos = n - pad + k d, one phase
//   ConvTranspose1d (stride s, K = m s)  output t = s q + r - pad takes taps k = r + j s of inputs q - j: s PHASES r, each a
//                                        m-tap convolution over q with its own weight slice, written with stride s
//   M = 96 (or 64) output channels per workgroup, N = 256 positions (4 waves x 64), K-loop = input channels in chunks of 32.
//   x is read as float32 [B][Ci][T] (the codec's layout, already Snake-activated by its producer), split and TRANSPOSED into LDS
//   rows [position][32 channels] (80-byte pitch: 16 lanes x 16 B cover all 64 banks), so a lane's B operand (8 consecutive input
//   channels of one position) is one ds_read_b128.  Weights are pre-split and pre-swizzled at load time into the exact A-operand
//   lane order, one contiguous 1 KiB wave load per (tap, 16 channels, 32 outputs, hi|lo), read straight from L2 one step ahead.
//   Epilogue: bias, residual add, raw output and/or the NEXT layer's Snake activation (hardware sine), as in ndac.hip.
#include <math.h>
#include <string.h>

#include <vector>

#include "internal.h"

namespace {

constexpr int TN = 256;    // output positions per workgroup
constexpr int CK = 32;     // input channels per LDS chunk
constexpr int ROWB = 80;   // LDS row pitch in bytes (32 bf16 + 16 B pad)

struct MArgs {
  const float* x; const uint4* wp; const float* bias; const float* res; float* out; float* out_act; const float* alpha_out;
  int Ci, T, Co, To, N;      // N: GEMM columns per (batch item, phase)
  int ntaps, nphase;
  int xlo;                   // input position of LDS row 0 = n0 + xlo
  int rbase, rstep;          // LDS row of (local column c, tap j) = c + rbase + j * rstep
  int rows;                  // staged rows = TN + span
  int ostride, ooff;         // output position of column n, phase r = n * ostride + r + ooff
  int nchunk;                // K-loop chunks of 32 GEMM-K elements
  int pad;                   // strided form: input position of (row m, residue r) = S m + r - pad
};

__device__ __forceinline__ float snake_fast(float v, float alpha, float inv) {
  const float s = __sinf(alpha * v);
  return v + inv * (s * s);
}

struct Epi { int b, co0, tpos[2]; bool ok[2]; };   // ([2]: NT <= 2)

// 8 rows of a tile at a time; the residual values of the NEXT 8 rows are requested before the current rows are stored.  FULL (every
// column of the wave's tile is a real output): the stores are unconditional -- loads and stores return in order through one counter
// (vmcnt), and with control flow between them hipcc has to wait as if no store were in flight, i.e. for all of them (elementwise.hip,
// fir_raw).
template <int MT, int NT, bool RES, bool OUT, bool ACT, bool FULL>
__device__ __forceinline__ void epilogue(const MArgs& a, const Epi& ep, f32x16 (&acc)[MT][NT]) {
  constexpr int EB = 4;   // rows per batch
  float rn[EB][NT];
  auto request = [&](int mt, int eh) {
#pragma unroll
    for (int e = 0; e < EB; ++e) {
      const size_t rowo = ((size_t)ep.b * a.Co + ep.co0 + 32 * mt + 8 * ((eh + e) >> 2) + (e & 3)) * a.To;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) rn[e][nt] = a.res[rowo + ep.tpos[nt]];
    }
  };
  if (RES) request(0, 0);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int eh = 0; eh < 16; eh += EB) {
      float rv[EB][NT];
      if (RES) {
#pragma unroll
        for (int e = 0; e < EB; ++e)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) rv[e][nt] = rn[e][nt];
        if (eh + EB < 16) request(mt, eh + EB);
        else if (mt + 1 < MT) request(mt + 1, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int e = 0; e < EB; ++e) {
        const int co = ep.co0 + 32 * mt + 8 * ((eh + e) >> 2) + (e & 3);
        const float bv = a.bias[co];
        float ao = 0.f, inv = 0.f;
        if (ACT) { ao = a.alpha_out[co]; inv = __builtin_amdgcn_rcpf(ao + 1e-9f); }
        const size_t rowo = ((size_t)ep.b * a.Co + co) * a.To;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          float v = acc[mt][nt][eh + e] + bv;
          if (RES) v += rv[e][nt];
          if (FULL || ep.ok[nt]) {
            if (OUT) a.out[rowo + ep.tpos[nt]] = v;
            if (ACT) a.out_act[rowo + ep.tpos[nt]] = snake_fast(v, ao, inv);
          }
        }
      }
    }
}

template <int MT, int NT, bool FULL>
__device__ __forceinline__ void epilogue_dispatch(const MArgs& a, const Epi& ep, f32x16 (&acc)[MT][NT]) {
  if (a.res) {
    if (a.out) epilogue<MT, NT, true, true, true, FULL>(a, ep, acc);
    else epilogue<MT, NT, true, false, true, FULL>(a, ep, acc);
  } else if (a.out) {
    if (a.out_act) epilogue<MT, NT, false, true, true, FULL>(a, ep, acc);
    else epilogue<MT, NT, false, true, false, FULL>(a, ep, acc);
  } else {
    epilogue<MT, NT, false, false, true, FULL>(a, ep, acc);
  }
}

// S == 0: stride-1 convolution / transposed convolution (a chunk = 32 input channels of one position).
// S >= 2: strided convolution Conv1d(K = m S, stride S) in polyphase form: out[n] = sum_j sum_r W[.., r + j S] x[S (n + j) + r - pad];
//         an LDS row is output-rate position m, a chunk = floor(32 / S) input channels x S residues (GEMM-K element e = c S + r;
//         32 - S floor(32 / S) zero columns), taps j = 0 .. m - 1 read rows n + j.  The S residues of a channel are S consecutive
//         samples: the staging reads of neighbouring lanes are contiguous.
// NT: 32-column MFMA tiles per wave (2: 256 positions per workgroup; 1: 128 -- twice the workgroups for short sequences / one clip).
template <int MT, int S, int NT>
__global__ __launch_bounds__(256, 2) void conv1d_mfma_kernel(MArgs a) {
  constexpr int TNK = 128 * NT;
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned char* const xh = smem;
  unsigned char* const xl = smem + a.rows * ROWB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.z, n0 = blockIdx.x * TNK;
  const int ncob = a.Co / (32 * MT);
  const int vcb = blockIdx.y, phase = vcb / ncob, co0 = (vcb - phase * ncob) * 32 * MT;
  const int nchunk = a.nchunk;
  constexpr int STEP = MT * 2 * 64;   // uint4 per (chunk, tap, 16-channel block)
  const uint4* wq = a.wp + (size_t)vcb * nchunk * a.ntaps * 2 * STEP + lane;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;

  uint4 ah[MT], al[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) { ah[mt] = wq[(mt * 2) * 64]; al[mt] = wq[(mt * 2 + 1) * 64]; }

  const float* const xb = a.x + (size_t)b * a.Ci * a.T;
  const long long pos0 = (long long)n0 + a.xlo;
  const int col = wave * 32 * NT + (lane & 31);
  const int koff = (lane >> 5) * 16;

  if (S > 0) {   // the zero columns are written once
    constexpr int CPC = S > 0 ? CK / S : 1, USED = CPC * S;
    if (USED < CK)
      for (int p = tid; p < a.rows; p += 256)
        for (int e = USED; e < CK; ++e) {
          *reinterpret_cast<unsigned short*>(xh + p * ROWB + 2 * e) = 0;
          *reinterpret_cast<unsigned short*>(xl + p * ROWB + 2 * e) = 0;
        }
  }

  for (int chunk = 0; chunk < nchunk; ++chunk) {
    __syncthreads();
    if (S == 0) {
#pragma unroll 1
      // (16-byte loads along T with a register transposition -- 4 positions x 4 channels per item, unaligned dwordx4 -- measured slower:
      //  decode 10.8 -> 11.6 ms, one clip 3.5 -> 4.4 ms)
      for (int g = 0; g < 4; g += 2) {   // two 8-channel groups per pass: 16 loads in flight per thread
        const float* const src = xb + (size_t)(chunk * CK + 8 * g) * a.T;
        for (int p = tid; p < a.rows; p += 256) {
          const long long tp = pos0 + p;
          float v[16];
          if (tp >= 0 && tp < a.T) {
            const float* q = src + tp;
#pragma unroll
            for (int j = 0; j < 16; ++j) { v[j] = *q; q += a.T; }
          } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = 0.f;
          }
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            bf16x8 hi, lo;
#pragma unroll
            for (int j = 0; j < 8; ++j) { hi[j] = (bf16)v[8 * h + j]; lo[j] = (bf16)(v[8 * h + j] - (float)hi[j]); }
            *reinterpret_cast<bf16x8*>(xh + p * ROWB + (g + h) * 16) = hi;
            *reinterpret_cast<bf16x8*>(xl + p * ROWB + (g + h) * 16) = lo;
          }
        }
      }
    } else {
      constexpr int CPC = S > 0 ? CK / S : 1;
      // item = (row p, channel c of the chunk): S consecutive samples -> S consecutive bf16 columns
      for (int it = tid; it < a.rows * CPC; it += 256) {
        const int c = it / a.rows, p = it - c * a.rows;
        const int ci = chunk * CPC + c;
        const long long q0 = ((long long)n0 + p) * S - a.pad;
        const float* const src = xb + (size_t)(ci < a.Ci ? ci : 0) * a.T;
        float v[S > 0 ? S : 1];
#pragma unroll
        for (int r = 0; r < S; ++r) {
          const long long q = q0 + r;
          v[r] = ci < a.Ci && q >= 0 && q < a.T ? src[q] : 0.f;
        }
        unsigned char* const dh = xh + p * ROWB + 2 * c * S;
        unsigned char* const dl = xl + p * ROWB + 2 * c * S;
        if (S % 2 == 0) {
#pragma unroll
          for (int r = 0; r < S; r += 2) {
            bf16x2 hi, lo;
            hi[0] = (bf16)v[r]; hi[1] = (bf16)v[r + 1];
            lo[0] = (bf16)(v[r] - (float)hi[0]); lo[1] = (bf16)(v[r + 1] - (float)hi[1]);
            *reinterpret_cast<bf16x2*>(dh + 2 * r) = hi;
            *reinterpret_cast<bf16x2*>(dl + 2 * r) = lo;
          }
        } else {
#pragma unroll
          for (int r = 0; r < S; ++r) {
            const bf16 hi = (bf16)v[r];
            *reinterpret_cast<bf16*>(dh + 2 * r) = hi;
            *reinterpret_cast<bf16*>(dl + 2 * r) = (bf16)(v[r] - (float)hi);
          }
        }
      }
    }
    __syncthreads();
#pragma unroll 1
    for (int tap = 0; tap < a.ntaps; ++tap) {
      const int r0 = (col + a.rbase + tap * a.rstep) * ROWB + koff;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        wq += STEP;   // the packed buffer carries one step of padding: the last prefetch stays in bounds
        uint4 nh[MT], nl[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { nh[mt] = wq[(mt * 2) * 64]; nl[mt] = wq[(mt * 2 + 1) * 64]; }
        bf16x8 bh[NT], bl[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          bh[nt] = *reinterpret_cast<const bf16x8*>(xh + r0 + nt * 32 * ROWB + kb * 32);
          bl[nt] = *reinterpret_cast<const bf16x8*>(xl + r0 + nt * 32 * ROWB + kb * 32);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al[mt]), bh[nt], acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[mt]), bl[nt], acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[mt]), bh[nt], acc[mt][nt], 0, 0, 0);
          }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { ah[mt] = nh[mt]; al[mt] = nl[mt]; }
      }
    }
  }

  // epilogue: lane = column (lane & 31) of each 32-wide tile, rows 8 (e / 4) + 4 (lane / 32) + e % 4
  Epi ep;
  ep.b = b; ep.co0 = co0 + 4 * (lane >> 5);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const long long n = (long long)n0 + col + 32 * nt;
    const long long t = n * a.ostride + phase + a.ooff;
    ep.ok[nt] = n < a.N && t >= 0 && t < a.To;
    ep.tpos[nt] = ep.ok[nt] ? (int)t : 0;   // (clamped: the residual loads are unconditional, the stores predicated)
  }
  // uniform per wave: columns that are all real outputs take the branch-free epilogue
  bool full = true;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) full = full && ep.ok[nt];
  if (__builtin_amdgcn_ballot_w64(!full) == 0) epilogue_dispatch<MT, NT, true>(a, ep, acc);
  else epilogue_dispatch<MT, NT, false>(a, ep, acc);
}

unsigned short bf16_rne(float f) {
  unsigned u; memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
float bf16_f32(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

int block_mt(int Co) { return Co % 96 == 0 ? 3 : (Co % 64 == 0 ? 2 : 0); }

bool strided_form(int stride, int transposed) { return !transposed && stride > 1; }
bool stride_instantiated(int s) { return s == 2 || s == 4 || s == 5 || s == 8 || s == 10; }
int chunks_of(int Ci, int stride, int transposed) { return strided_form(stride, transposed) ? fd_cdiv(Ci, CK / stride) : Ci / CK; }

template <int MT, int NT>
int launch(const MArgs& a, int S, dim3 grid, size_t lds, hipStream_t st) {
  switch (S) {
    case 0: hipLaunchKernelGGL((conv1d_mfma_kernel<MT, 0, NT>), grid, dim3(256), lds, st, a); break;
    case 2: hipLaunchKernelGGL((conv1d_mfma_kernel<MT, 2, NT>), grid, dim3(256), lds, st, a); break;
    case 4: hipLaunchKernelGGL((conv1d_mfma_kernel<MT, 4, NT>), grid, dim3(256), lds, st, a); break;
    case 5: hipLaunchKernelGGL((conv1d_mfma_kernel<MT, 5, NT>), grid, dim3(256), lds, st, a); break;
    case 8: hipLaunchKernelGGL((conv1d_mfma_kernel<MT, 8, NT>), grid, dim3(256), lds, st, a); break;
    case 10: hipLaunchKernelGGL((conv1d_mfma_kernel<MT, 10, NT>), grid, dim3(256), lds, st, a); break;
    default: return fd_set_error(FD_EINVAL, "ndac mfma conv: stride %d not instantiated", S);
  }
  FD_LAUNCH_CHECK();
  return FD_OK;
}

}  // namespace

bool fd_ndac_mfma_supported(int Ci, int Co, int K, int stride, int dil, int transposed) {
  if (Ci <= 0 || !block_mt(Co) || K <= 0 || stride <= 0 || dil <= 0) return false;
  if (strided_form(stride, transposed)) {
    if (!stride_instantiated(stride) || K % stride != 0 || dil != 1) return false;
  } else {
    if (Ci % CK) return false;
    if (transposed ? (K % stride != 0 || dil != 1) : stride != 1) return false;
  }
  const int span = stride > 1 ? K / stride - 1 : (K - 1) * dil;
  return (size_t)(TN + span) * ROWB * 2 <= 64 * 1024;
}

size_t fd_ndac_mfma_packed_bytes(int Ci, int Co, int K, int stride, int transposed) {   // + one step of prefetch padding
  const int ntaps = stride > 1 ? K / stride : K, nphase = transposed ? stride : 1;
  const size_t steps = (size_t)nphase * chunks_of(Ci, stride, transposed) * ntaps * 2;
  return (steps * Co * 16 * 2 + 3 * 2 * 64 * 8) * sizeof(unsigned short);
}

// w: [Ci][K][Co] float32 (the codec's own layout for all three kinds) -> A-operand order
//   [phase * Co / CB + co block][chunk of 32 GEMM-K elements][tap][16-element block][32-co tile][hi | lo][lane][8]
// GEMM-K element e of chunk c:  stride-1 / transposed: input channel 32 c + e;  strided: channel c floor(32 / S) + e / S, residue e % S
// (kernel tap r + j S), zero beyond floor(32 / S) S or Ci.
void fd_ndac_mfma_pack(const float* w, int Ci, int Co, int K, int stride, int transposed, void* dst) {
  const int mt_n = block_mt(Co), CB = 32 * mt_n, ncob = Co / CB;
  const bool str = strided_form(stride, transposed);
  const int nphase = transposed ? stride : 1, ntaps = stride > 1 ? K / stride : K, nchunk = chunks_of(Ci, stride, transposed);
  const int cpc = str ? CK / stride : CK;
  unsigned short* o = static_cast<unsigned short*>(dst);
  memset(o, 0, fd_ndac_mfma_packed_bytes(Ci, Co, K, stride, transposed));
  for (int ph = 0; ph < nphase; ++ph)
    for (int cob = 0; cob < ncob; ++cob)
      for (int chunk = 0; chunk < nchunk; ++chunk)
        for (int tap = 0; tap < ntaps; ++tap)
          for (int kb = 0; kb < 2; ++kb)
            for (int mt = 0; mt < mt_n; ++mt)
              for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                  const int co = cob * CB + 32 * mt + (lane & 31), e = kb * 16 + (lane >> 5) * 8 + j;
                  int ci, k;
                  if (str) {
                    if (e >= cpc * stride) continue;
                    ci = chunk * cpc + e / stride; k = e % stride + tap * stride;
                    if (ci >= Ci) continue;
                  } else {
                    ci = chunk * CK + e; k = transposed ? ph + tap * stride : tap;
                  }
                  const float v = w[((size_t)ci * K + k) * Co + co];
                  const unsigned short h = bf16_rne(v), l = bf16_rne(v - bf16_f32(h));
                  const size_t step = (((size_t)(ph * ncob + cob) * nchunk + chunk) * ntaps + tap) * 2 + kb;
                  const size_t base = ((step * mt_n + mt) * 2) * 64 * 8 + (size_t)lane * 8 + j;
                  o[base] = h;
                  o[base + 64 * 8] = l;
                }
}

int fd_ndac_mfma_conv(const float* x, const void* wp, const float* bias, const float* res, float* out, float* out_act, const float* alpha_out, int B,
                      int Ci, int T, int Co, int K, int stride, int pad, int dil, int transposed, hipStream_t st) {
  FD_REQUIRE(fd_ndac_mfma_supported(Ci, Co, K, stride, dil, transposed), "ndac mfma conv: unsupported shape Ci %d Co %d K %d stride %d dilation %d", Ci, Co, K,
             stride, dil);
  FD_REQUIRE(x && wp && bias && (out || out_act) && (!out_act || alpha_out), "ndac mfma conv: null argument");
  MArgs a{};
  a.x = x; a.wp = static_cast<const uint4*>(wp); a.bias = bias; a.res = res; a.out = out; a.out_act = out_act; a.alpha_out = alpha_out;
  a.Ci = Ci; a.T = T; a.Co = Co; a.pad = pad;
  a.nchunk = chunks_of(Ci, stride, transposed);
  long long To;
  int S = 0, span = 0;
  if (transposed) {
    a.ntaps = K / stride; a.nphase = stride;
    To = ((long long)T - 1) * stride - 2 * pad + K;
    a.N = T + a.ntaps - 1;
    a.xlo = -(a.ntaps - 1); a.rbase = a.ntaps - 1; a.rstep = -1; span = a.ntaps - 1;
    a.ostride = stride; a.ooff = -pad;
  } else if (stride > 1) {
    S = stride;
    a.ntaps = K / stride; a.nphase = 1;
    To = ((long long)T + 2 * pad - K) / stride + 1;
    a.N = (int)To;
    a.xlo = 0; a.rbase = 0; a.rstep = 1; span = a.ntaps - 1;
    a.ostride = 1; a.ooff = 0;
  } else {
    a.ntaps = K; a.nphase = 1;
    To = (long long)T + 2 * pad - (long long)dil * (K - 1);
    a.N = (int)To;
    a.xlo = -pad; a.rbase = 0; a.rstep = dil; span = (K - 1) * dil;
    a.ostride = 1; a.ooff = 0;
  }
  FD_REQUIRE(To > 0 && To < (1ll << 31), "ndac mfma conv: empty output");
  a.To = (int)To;
  const int mt = block_mt(Co);
  // 256 positions per workgroup unless that leaves fewer than two workgroups per CU (short sequences, one clip): then 128
  const long long wgs256 = (long long)fd_cdiv(a.N, TN) * (Co / (32 * mt)) * a.nphase * B;
  const int nt = wgs256 >= 512 ? 2 : 1, tn = 128 * nt;
  a.rows = tn + span;
  const dim3 grid(fd_cdiv(a.N, tn), Co / (32 * mt) * a.nphase, B);
  const size_t lds = (size_t)a.rows * ROWB * 2;
  if (nt == 2) return mt == 3 ? launch<3, 2>(a, S, grid, lds, st) : launch<2, 2>(a, S, grid, lds, st);
  return mt == 3 ? launch<3, 1>(a, S, grid, lds, st) : launch<2, 1>(a, S, grid, lds, st);
}
