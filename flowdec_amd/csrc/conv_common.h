// conv_common.h -- definitions shared by the two MFMA convolution kernels (conv_mfma.hip: direct implicit GEMM,
// conv_wino.hip: Winograd F(2,3) along W) of libflowdec_hip.so.
#pragma once
#include "common.h"
#include "internal.h"

// Per-workgroup phase timing (s_memtime stamps written to the buffer of fd_debug_buffer) exists in -DFD_TIMING2 builds only
// (scripts/build_t2.sh): FD_T2(...) keeps its argument there and drops it otherwise.
#ifdef FD_TIMING2
#define FD_T2(...) __VA_ARGS__
#else
#define FD_T2(...)
#endif

namespace fdconv {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int ROWB = 80;   // halo row pitch in LDS (64 B of data + 16 B pad: conflict-free b128 reads at any tap offset)
constexpr int WROWB = 64;  // weight row pitch, global (packed) and LDS: no padding -- the four 16-B columns of row r are
                           // stored XOR-swizzled by (r >> 2) & 3, which makes the b128 fragment reads conflict-free too
                           // and keeps the LDS-DMA traffic at exactly the useful bytes
constexpr int PITCH = 24;  // halo row pitch in pixels

struct Seg {
  const void* src;
  int C;         // channels of this tensor
  int aff_off;   // channel offset into the affine table, or -1 (no activation)
  int taps;      // 9 or 1
};

struct ConvArgs {
  Seg seg[4];
  int nseg;
  const float* affine; int affC;   // [B][affC][2]
  const void* w;                   // packed [step][CoutPad][WROWB bytes]
  long long w_bytes;
  const float* w_scale;            // Winograd kernels: [CoutPad] f32, inverse of the per-cout power-of-two scale of the packed weights
  const float* bias; int bias_rows;
  const void* skip;
  float scale;
  void* out;
  int Cout, CoutPad;
  float* stats;                    // [B][tiles_h*tiles_w][CoutPad][2] or null
  unsigned long long* dbg;         // FD_TIMING2 builds only
  int B, H, W;
  int tiles_h, tiles_w, tiles_n;
  int reversed;                    // FD_TILE_REVERSED: workgroup i takes tile (n - 1 - i)
};


constexpr int AFF_BYTES = 512 * 8;  // per-(b,c) (a,d) pairs of up to 512 activated input channels, staged in LDS

}  // namespace fdconv

// conv_wino.hip
long long fd_wino_packed_bytes(int Cout, int C0, int C1, int S0, int S1);
int fd_wino_pack_weights(const float* w, const float* w_sc, void* packed, int Cout, int C0, int C1, int S0, int S1, hipStream_t st);
int fd_wino_launch(fdconv::ConvArgs a, hipStream_t st);
int fd_wino_init_attributes();
bool fd_wino_supported(int Cout, int C0, int C1, int S0, int S1, int ksize);

// conv_wino4.hip (Winograd F(4,3) along W, 256-cout workgroups; bf16 storage, whole 16 x 16 tiles)
long long fd_wino4_packed_bytes(int Cout, int C0, int C1, int S0, int S1);
int fd_wino4_pack_weights(const float* w, const float* w_sc, void* packed, int Cout, int C0, int C1, int S0, int S1, hipStream_t st);
int fd_wino4_launch(fdconv::ConvArgs a, hipStream_t st);
int fd_wino4_init_attributes();
bool fd_wino4_supported(int Cout, int C0, int C1, int S0, int S1, int ksize);
bool fd_wino4_shape_ok(int H, int W);

// conv_wino4f.hip (the same algorithm in exact float32: f32 storage, v_mfma_f32_32x32x2_f32; channel counts % 16 == 0)
long long fd_wino4f_packed_bytes(int Cout, int C0, int C1, int S0, int S1);
int fd_wino4f_pack_weights(const float* w, const float* w_sc, void* packed, int Cout, int C0, int C1, int S0, int S1, hipStream_t st);
int fd_wino4f_launch(fdconv::ConvArgs a, hipStream_t st);
int fd_wino4f_init_attributes();
bool fd_wino4f_supported(int Cout, int C0, int C1, int S0, int S1, int ksize);
bool fd_wino4f_shape_ok(int H, int W);

// conv_wino44f.hip (2-D Winograd F(4x4, 3x3) in exact float32: 128-cout workgroups, whole 16 x 16 tiles, channel counts % 8 == 0)
long long fd_wino44f_packed_bytes(int Cout, int C0, int C1, int S0, int S1);
int fd_wino44f_pack_weights(const float* w, const float* w_sc, void* packed, int Cout, int C0, int C1, int S0, int S1, hipStream_t st);
int fd_wino44f_launch(fdconv::ConvArgs a, hipStream_t st);
int fd_wino44f_init_attributes();
bool fd_wino44f_supported(int Cout, int C0, int C1, int S0, int S1, int ksize);
bool fd_wino44f_shape_ok(int H, int W);

// conv_head.hip (Cout = 4 pyramid heads, bf16)
bool fd_head_supported(const fdconv::ConvArgs& a, int ksize, int dtype);
int fd_head_launch(fdconv::ConvArgs a, hipStream_t st);
int fd_head_init_attributes();

// conv_headf.hip (Cout = 4 pyramid heads, exact float32: v_mfma_f32_4x4x1_16B_f32)
bool fd_headf_supported(const fdconv::ConvArgs& a, int ksize, int dtype);
int fd_headf_launch(fdconv::ConvArgs a, hipStream_t st);
int fd_headf_init_attributes();
