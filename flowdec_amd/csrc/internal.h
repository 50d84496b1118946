// internal.h -- cross-file (non-ABI) declarations of libflowdec_hip.so.
#pragma once
#include "common.h"

struct fd_temb_job {
  const float* dense_w;  // [Cout][temb_dim]
  const float* dense_b;  // [Cout]
  const float* conv_b;   // [Cout]
  float* out;            // [nt][Cout]
  int Cout;
};

struct fd_edge_args {
  const void* x = nullptr;   // main input (4-channel NHWC tensor, or complex x for pack)
  const void* y = nullptr;   // second input (complex y for pack; h for combine)
  const float* w = nullptr;
  const float* bias = nullptr;
  void* out = nullptr;
  const void* base = nullptr;  // output_update
  const void* kold = nullptr;
  void* ksave = nullptr;
  float* stats = nullptr;      // combine: [B][fd_combine_tiles][Cout][2] GroupNorm partials of the output
  float coef = 1.f;
  const void* z = nullptr;     // score update: dst = cb*base + cy*y + coef*v + cz*z
  float cb = 1.f, cy = 0.f, cz = 0.f;
  int B = 0, H = 0, W = 0, Cout = 0;
};

// elementwise.hip
int fd_time_embedding_impl(const float* t, float t_imm, int nt, const float* gfp_w, int nf, const float* w1, const float* b1,
                           const float* w2, const float* b2, float* temb, hipStream_t st);
int fd_temb_bias_batched(const fd_temb_job* jobs_dev, int njobs, const float* temb, int nt, int temb_dim, hipStream_t st);
// which: 0 = pack_input, 2 = combine (1x1 4->Cout + h), 3 = output layer + state update,
//        4 = output layer + score-sampler update, 5 = input convolution 3x3 4 -> Cout (x = packed input, w = [Cout][4][3][3] f32) with
//        the GroupNorm partials of its output in `stats` ([B][(H / 16) * (W / 16)][Cout][2])
int fd_edge_op(int which, const fd_edge_args& a, int dtype, hipStream_t st);
int fd_init_state(const float* Y, const float* noise, const double* sigma_dev, int sigma_n, float sigma_fac, float* x0, int B,
                  int F, int T, hipStream_t st);
int fd_combine_tiles(int H, int W);
// adaptive solver helpers: dst = cx * x + dt * sum c[i] k[i];  partial[b] = sum |p - q|^2 / (atol + rtol max(|r|, |s|))^2
int fd_ode_lincomb(const float* x, float cx, float dt, const float* const* k, const float* c, int nk, float* dst, long long n, hipStream_t st);
int fd_ode_scaled_sq(const float* p, const float* q, const float* r, const float* s, float atol, float rtol, double* partial, int nblocks,
                     long long n, hipStream_t st);
int fd_caxpy(const float* a, const float* q, float cq, float* dst, long long n, hipStream_t st);
// conv_mfma.hip
int fd_conv_init_attributes();
// stft.hip
// (fd_stft_plan / fd_stft_plan_create / fd_stft_plan_destroy: public, include/flowdec_hip.h)
// lens: device int32 [B] per-clip sample counts of a ragged batch, or nullptr (every clip is L samples long)
int fd_stft_forward(fd_stft_plan* p, const float* y, const int* lens, int B, int L, float alpha, float beta, int normalize, float* normfac,
                    float* Y, int T_pad, void* ws, size_t ws_bytes, hipStream_t st);
int fd_stft_inverse(fd_stft_plan* p, const float* X, const int* lens, int B, int T, int T_pad, float alpha, float beta, const float* normfac,
                    float* y, int L, void* ws, size_t ws_bytes, hipStream_t st);
size_t fd_stft_ws_bytes(int B, int L, int n_fft, int hop);
// ndac_mfma.hip: the codec's wide convolutions on the matrix cores (split-bf16 operands, f32 tolerance)
bool fd_ndac_mfma_supported(int Ci, int Co, int K, int stride, int dil, int transposed);
size_t fd_ndac_mfma_packed_bytes(int Ci, int Co, int K, int stride, int transposed);
void fd_ndac_mfma_pack(const float* w_ci_k_co, int Ci, int Co, int K, int stride, int transposed, void* dst);
int fd_ndac_mfma_conv(const float* x, const void* wp, const float* bias, const float* res, float* out, float* out_act, const float* alpha_out, int B,
                      int Ci, int T, int Co, int K, int stride, int pad, int dil, int transposed, hipStream_t st);
