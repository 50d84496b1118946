/* flowdec_hip.h -- C ABI of libflowdec_hip.so: the MI355X (gfx950) drop-in for the FlowDec
 * inference hot path (FlowModel.enhance: STFT -> NCSN++ x NFE inside a fixed-step ODE loop ->
 * iSTFT).  Everything here is `extern "C"`, plain pointers and sizes; no torch types.
 *
 * Conventions
 *  - Every function returns 0 on success or a negative FD_E* code; fd_last_error() gives the
 *    message of the last failure on the calling thread.  The reference's native ops raise a C++
 *    exception -> Python RuntimeError (op/upfirdn2d.cpp:34-42); the Python binding in
 *    flowdec_amd/_lib.py turns a non-zero return into RuntimeError to keep that behaviour.
 *  - All data pointers are DEVICE pointers owned by the caller unless stated otherwise; nothing on
 *    the hot path allocates.  All work is enqueued asynchronously on `stream` (a hipStream_t passed
 *    as void*; NULL = default stream), no hidden synchronisation, graph-capture safe -- the same
 *    contract as the reference's launches on at::cuda::getCurrentCUDAStream (upfirdn2d_kernel.cu:224-226).
 *  - Activation tensors are NHWC ([B][H][W][C], H = frequency bins, W = time frames) in the storage
 *    type given by `dtype` (FD_F32 or FD_BF16).  Spectrogram / ODE-state tensors at the model boundary
 *    use the reference layout: complex64 [B][1][F=768][T] (interleaved re,im), float32 waveforms [B][L].
 *  - Threading.  The operator-level calls (fd_upfirdn2d, fd_fused_bias_act, fd_conv2d, fd_fir_resample, fd_gn_*, fd_stft_*, ...)
 *    keep no state and are re-entrant from any number of threads, like the reference's ops (upfirdn2d_kernel.cu:224-231).  An
 *    fd_model holds scratch that its enqueueing calls share (time-embedding biases, hipGraph cache, side stream, profiling events):
 *    it serves ONE enqueueing call at a time.  A second thread that enters fd_ncsnpp_forward / fd_ode_solve[_adaptive] / fd_enhance /
 *    fd_score_* / fd_regression_enhance while another is inside gets FD_EBUSY (nothing is enqueued, nothing is corrupted); use one
 *    fd_model per thread for concurrent solves.  "Inside" is the host-side enqueue only -- the GPU work itself is asynchronous:
 *    consecutive calls of one model on DIFFERENT streams must be ordered by the caller (record an event after one call, make the
 *    next stream wait for it), because the workspace passed in and the model's own scratch (the per-step time-embedding biases)
 *    are reused from call to call.  The Python binding does both for its callers (flowdec_amd/model.py: _NativeCall).
 */
#ifndef FLOWDEC_HIP_H
#define FLOWDEC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FD_OK 0
#define FD_EINVAL (-1)   /* bad argument / unsupported shape */
#define FD_ERUNTIME (-2) /* HIP runtime error (message has hipGetErrorString) */
#define FD_ENOMEM (-3)   /* caller-provided workspace too small */
#define FD_ESTATE (-4)   /* model not finalised / parameter missing */
#define FD_EBUSY (-5)    /* the fd_model is inside an enqueueing call of another thread (see "Threading" below) */

#define FD_F32 0
#define FD_BF16 1
/* Algorithm flag, OR-ed into the `dtype` / `wdtype` / `act_dtype` arguments that select a convolution (storage type =
 * value & 0xff): 3x3 convolutions with Cout % 128 == 0 and all channel counts % 32 == 0 run as Winograd F(2,3) along W
 * (1.5x fewer MFMAs; bf16 storage, fp16 MFMA operands, f32 accumulation; conv_wino.hip).  The packed weights of the two
 * algorithms differ: pack and launch with the same flag.  Both Winograd packings scale every output channel's transformed weights
 * (and its folded shortcut weights) by a power of two before the fp16 rounding and carry the inverse table behind the packed steps;
 * the kernels undo it exactly in the epilogue -- weights of any magnitude (1e-6 .. 1e2 tested, per channel) keep their mantissa. */
#define FD_WINOGRAD 0x100
/* Same places as FD_WINOGRAD: 3x3 convolutions with Cout == 256, all channel counts % 32 == 0 and whole 16 x 16 pixel tiles
 * (H % 16 == W % 16 == 0) as Winograd F(4,3) along W -- HALF the MFMAs of the direct kernel; 256-cout workgroups with the input
 * transform done once per workgroup when the activated halo is stored (conv_wino4.hip).  bf16 storage, fp16 MFMA operands, f32
 * accumulation; 3x3 inputs, raw or activated, saturate at +-6000 in their fp16 form (a folded shortcut input never passes through
 * fp16).  Pack and launch with the same flag. */
#define FD_WINOGRAD4 0x80000
/* fd_conv2d with FD_WINOGRAD4 only: walk the pixel tiles in DESCENDING order.  Same result, bit for bit; a consumer that starts where
 * its producer stopped finds the producer's last output lines still in the memory-side cache (the model alternates the direction from
 * one F(4,3) launch to the next: 0.5 % of a cfg 2 step). */
/* fd_conv_pack_weights / fd_conv2d with FD_F32 storage: 2-D Winograd F(4x4, 3x3) in exact float32 (conv_wino44f.hip: 2.25 multiply-adds per
 * output and input channel instead of 9; Cout % 128 == 0, channel counts % 8 == 0, H % 16 == W % 16 == 0; folded shortcut and residual
 * input allowed together).  The fp32 mode's kernel (`FD_F32 | FD_WINOGRAD_AUTO`). */
#define FD_WINOGRAD44 0x200000
#define FD_TILE_REVERSED 0x100000
/* fd_model_config.act_dtype only: Winograd for the blocks of resolution level >= 2 (small grids, where its 128-cout workgroups
 * fill the chip better), direct MFMA convolution elsewhere. */
#define FD_WINOGRAD_LOWRES 0x200
/* fd_model_config.act_dtype only: all packings are kept and every launch picks its kernel by the image size: the direct kernel with
 * FD_TILE_BN64_CHUNK for images of at most 16 tiles of 16 x 16 pixels, Winograd F(2,3) up to 96 tiles (the low-resolution levels;
 * not with a folded 1x1 shortcut), Winograd F(4,3) for every other 3x3 convolution with 256 output channels and whole tiles (every
 * other such launch of a forward with FD_TILE_REVERSED), direct for the rest.  Independent of the batch size, so that a clip gives the
 * same bits alone and inside any batch. */
#define FD_WINOGRAD_AUTO 0x400
/* fd_conv_pack_weights / fd_conv2d / fd_model_config.act_dtype, with FD_F32 only: "bf16 operands, f32 residual stream" -- activations,
 * skip tensors and outputs stay f32 in memory; a convolution rounds its (activated) input to bf16 at the LDS store, its weights are
 * packed as bf16, accumulation is f32.  Direct kernel, default workgroup widths. */
#define FD_BF16_OPERANDS 0x10000
/* same places, with FD_F32 only: every conv operand as the two-term bf16 split x = hi + lo (16 mantissa bits) and every product as
 * hi*hi + hi*lo + lo*hi on the bf16 matrix cores, f32 accumulation -- results within the f32 mode's tolerances (a conv is ~1e-5 from
 * the f64 convolution) at 3x the bf16 MFMA work instead of the f32 MFMA's 16x. */
#define FD_BF16X3_OPERANDS 0x20000
/* fd_conv2d only (direct kernel): output channels per workgroup, 32 / 64 / 128 instead of the default min(256, padded Cout).  Narrow
 * workgroups put a SMALL image on more compute units (latency) at the price of re-activating the input once per workgroup
 * (throughput).  The convolution result is bit-identical for every width (same K order per output); the per-tile statistics
 * differ in summation order only. */
#define FD_TILE_BN32 0x1000
#define FD_TILE_BN64 0x2000
#define FD_TILE_BN128 0x3000
#define FD_TILE_BN64_CHUNK 0x4000 /* 64 + the weight slabs of two whole 32-channel chunks resident in LDS: one barrier and one memory
                                     round trip per chunk instead of per tap pair (bf16; falls back to FD_TILE_BN64 otherwise) */
#define FD_TILE_BN32_CHUNK 0x5000 /* same with 32-channel workgroups (4 waves) */
#define FD_TILE_DUO128 0x6000 /* 4 waves x (128 px x 64 cout) with ONE halo buffer: 70 KiB of LDS, two workgroups per CU (bf16, Cout % 128 == 0) */
#define FD_TILE_PERSIST 0x7000 /* "register epilogue, continuous tiles" (bf16, Cout == 128 or 256, H % 16 == W % 16 == 0, no residual input; the
                                  default configuration otherwise): one persistent workgroup per compute unit walks a contiguous range of tiles
                                  as ONE software pipeline, epilogue on the accumulator registers, stores left in flight.  Bit-identical
                                  convolution result.  Measured (profiles/r03_register_epilogue.txt): no prologue, 12 % fewer instructions
                                  around the MFMAs -- and 0.97-1.06x of the default: the K loop loses what the tile boundary gains.  Opt-in. */
#define FD_TILE_MASK 0xf000
/* fd_model_config.act_dtype only: low-latency schedule for ONE short clip -- both packings are kept (as with FD_WINOGRAD_AUTO) and
 * every convolution picks kernel and workgroup width by its IMAGE size (never by the batch size): FD_TILE_BN32_CHUNK for images of
 * at most 24 tiles, Winograd F(2,3) up to 128 tiles (unless a 1x1 shortcut is folded in: direct with FD_TILE_BN128 workgroups there),
 * Winograd F(4,3) above 128 tiles. */
#define FD_LOW_LATENCY 0x800
/* fd_model_config.act_dtype only: keep the side branches of a network evaluation (time embedding, pyramid-head chain) on the caller's
 * stream instead of forking them onto the model's second stream (default: forked; inside a graph capture they become parallel
 * branches of the graph).  Results are bit-identical either way. */
#define FD_NO_SIDE_STREAM 0x40000

/* solver ids (flowdec/model.py:487 'euler'/'midpoint' via torchdyn; sampling/solvers.py:15-57) */
#define FD_SOLVER_EULER 0
#define FD_SOLVER_MIDPOINT 1
#define FD_SOLVER_HEUN2 2
#define FD_SOLVER_HEUN2_EULERLAST 3

const char* fd_last_error(void);
int fd_version(void);
/* Box calibration (bench.py `box_calibration`): `repeats` timed launches of a fixed matrix-core issue loop (register-resident random
 * bf16 operands, no LDS, no memory; iters <= 0: 100000 iterations, ~50 ms per launch) on `stream`; *tflops = the mean rate the box
 * sustains at its power-limited clock, *ms_total (optional) the time of the timed launches.  scratch: 2 KiB of device memory.
 * Synchronises the stream.  No counterpart in the reference (enhance.py:120-136 times one call with two events). */
int fd_calibrate_mfma(float* scratch, int iters, int repeats, double* tflops, double* ms_total, void* stream);
/* Device properties the bench reports: [0]=CU count, [1]=max clock kHz, [2]=wavefront size, [3]=gfx arch number. */
int fd_device_info(int* out4);

/* ------------------------------------------------------------------------------------------------
 * Native-operator parity (the reference's only FFI)
 * ---------------------------------------------------------------------------------------------- */

/* Replaces upfirdn2d_op.upfirdn2d(input[major,in_h,in_w,minor], kernel[kh,kw], up_x,up_y,down_x,down_y,
 * pad_x0,pad_x1,pad_y0,pad_y1) -> out[major,out_h,out_w,minor]
 * (op/upfirdn2d.cpp:38-49; kernel upfirdn2d_kernel.cu:118-218; out_h/out_w formula :248-251).
 * `kernel` is float32 on device.  `out` must hold major*out_h*out_w*minor elements. */
int fd_upfirdn2d(const void* input, const float* kernel, void* out, int major, int in_h, int in_w, int minor,
                 int kernel_h, int kernel_w, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1,
                 int pad_y0, int pad_y1, int dtype, void* stream);
int fd_upfirdn2d_out_size(int in_size, int up, int down, int pad0, int pad1, int ksize);

/* Replaces fused_bias_act(input, bias, refer, act, grad, alpha, scale) for grad == 0
 * (op/fused_bias_act.cpp:37-46; kernel fused_bias_act_kernel.cu:30-61):
 * out[i] = scale * act(x[i] + bias[(i / step_b) % size_b]); act 1 = linear, 3 = leaky-relu(alpha).
 * bias may be NULL (size_b == 0).  float32 only (dead code on the hot path; kept for API parity). */
int fd_fused_bias_act(const float* x, const float* bias, float* out, long long n, int step_b, int size_b, int act,
                      float alpha, float scale, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Hot-path building blocks (NHWC activations)
 * ---------------------------------------------------------------------------------------------- */

/* StyleGAN2 FIR [1,3,3,1] x2 resampling in polyphase form (upsample_2d / downsample_2d,
 * up_or_down_sampling.py:220-282) on an NHWC tensor.  If `affine` != NULL ([B][C] pairs (a,d)) a second
 * output out_act = FIR(silu(a*x+d)) is produced from the same read (the resample of
 * h = act(GroupNorm_0(x)) and of x in ResnetBlockBigGANpp.forward, layerspp.py:255-267).
 * direction: +1 = up x2, -1 = down x2.  Either output may be NULL. */
int fd_fir_resample(const void* x, const float* affine, void* out_raw, void* out_act, int B, int H, int W, int C,
                    int direction, int dtype, void* stream);

/* The input convolution of NCSN++ (all_modules.3 = conv3x3(4, nf), ncsnpp.py:291 via layers.py:128-134) on the packed NHWC input
 * [B][H][W][8] (channels 0..3 = x.re, x.im, y.re, y.im; 4..7 ignored), zero padding, as f32 vector FMAs (taps ascending, input channels
 * ascending, one fma each), with the GroupNorm partial sums of the output: stats[B][(H / 16) * (W / 16)][Cout][2] = per 16 x 16 pixel
 * tile (sum x, sum x^2) of the f32 values.  w = [Cout][4][3][3] float32 (the checkpoint's layout), Cout in {8, 16, 32, 64},
 * H % 16 == W % 16 == 0.  dtype = storage type of `in8` and `out`. */
int fd_conv_in(const void* in8, const float* w, const float* bias, void* out, float* stats, int B, int H, int W, int Cout, int dtype,
               void* stream);

/* GroupNorm statistics (nn.GroupNorm(min(C//4,32), C, eps=1e-6), layerspp.py:229,241), split in two so that one
 * statistics pass can be shared by consumers that group the channels differently:
 *  partial sums  : part[b][tile][stride][2] = per-channel (sum x, sum x^2) of one spatial tile, float32.  Produced
 *                  either by fd_channel_sums (stand-alone pass; tiles = fd_channel_sums_tiles(H, W), stride = C) or by
 *                  fd_conv2d for its OUTPUT (tiles = fd_conv_stats_tiles(H, W), stride = fd_conv_cout_pad(Cout)).
 *  fd_gn_finalize: reduce the partials of one or two tensors (virtual channel concat [C0 | C1], ncsnpp.py:337) in
 *                  float64 and emit per-(b,c) affine pairs a = rstd*gamma[c], d = beta[c] - mean*rstd*gamma[c]. */
int fd_channel_sums_tiles(int H, int W);
int fd_channel_sums(const void* x, float* part, int B, int H, int W, int C, int dtype, void* stream);
int fd_gn_finalize(const float* part0, int tiles0, int stride0, int C0, const float* part1, int tiles1, int stride1, int C1,
                   const float* gamma, const float* beta, float* affine, int B, int groups, long long hw, float eps,
                   void* stream);

/* Packs PyTorch conv weights [Cout][Cin][k][k] float32 (device) into the MFMA layout [step][CoutPad][64 B]
 * (bf16: 32 channels, f32: 16 channels per row, the four 16-byte columns XOR-swizzled by (cout >> 2) & 3;
 * step = (concat segment, channel chunk, tap)).  Input channels
 * are split at C0 into two chunk-padded segments (virtual concat).  `w_sc` (optional, [Cout][S0+S1][1][1]) is the
 * 1x1 shortcut conv of a ResnetBlock (Conv_2, layerspp.py:244-245) whose K steps are appended so that one launch
 * computes Conv_1(h) + Conv_2(x). */
long long fd_conv_packed_bytes(int Cout, int C0, int C1, int ksize, int S0, int S1, int wdtype);
int fd_conv_pack_weights(const float* w, const float* w_sc, void* packed, int Cout, int C0, int C1, int ksize, int S0, int S1,
                         int wdtype, void* stream);
int fd_conv_cout_pad(int Cout);          /* row count of a packed slab / channel stride of the stats partials */
int fd_conv_stats_tiles(int H, int W);   /* 16x16 tiles per image */

/* Implicit-GEMM convolution, stride 1, 'same' zero padding, ksize 3 or 1 (ddpm_conv3x3 / ddpm_conv1x1,
 * layers.py:110-134) on MFMA:
 *   out = scale * ( conv_k( act([in0 | in1]) ) + conv_1x1([sc0 | sc1]) + bias[b] + skip )
 * act(x) = silu(a*x+d) per (b,c) if `affine` != NULL (GroupNorm+SiLU folded into the operand load), identity otherwise;
 * [in0|in1] / [sc0|sc1] are virtual channel concats (second tensors optional); bias: [bias_rows][Cout] float32 with
 * bias_rows in {1, B} (conv bias + Dense_0(act(temb)), layerspp.py:272-273); skip: optional NHWC tensor of Cout
 * channels ((x+h)/sqrt(2), layerspp.py:281-284); stats: optional partial sums of the output (see above).
 * dtype selects storage AND arithmetic: FD_BF16 = bf16 operands / f32 accumulate (v_mfma_f32_32x32x16_bf16),
 * FD_F32 = exact f32 (v_mfma_f32_32x32x2_f32).  Input channel counts must be multiples of 8, Cout 4 or a multiple of 8. */
int fd_conv2d(const void* in0, int C0, const void* in1, int C1, const float* affine, const void* sc0, int S0, const void* sc1,
              int S1, const void* packed_w, const float* bias, int bias_rows, const void* skip, float scale, void* out,
              int Cout, float* stats, int B, int H, int W, int ksize, int dtype, void* stream);

/* Time embedding: GaussianFourierProjection -> Linear -> SiLU -> Linear (ncsnpp.py:263-274,
 * layerspp.py:42-51); t [nt] float32 -> temb [nt][4*nf]. */
int fd_time_embedding(const float* t, int nt, const float* gfp_w, int nf, const float* w1, const float* b1,
                      const float* w2, const float* b2, float* temb, void* stream);
/* out[r][o] = conv_bias[o] + dense_b[o] + sum_k dense_w[o][k] * silu(temb[r][k])  (layerspp.py:272-273). */
int fd_temb_bias(const float* temb, int nt, int temb_dim, const float* dense_w, const float* dense_b,
                 const float* conv_bias, int Cout, float* out, void* stream);

/* silu(a*x + d) with per-(b,c) affine pairs = act(GroupNorm(x)) as a stand-alone pass (layerspp.py:253,274); on the hot
 * path this is fused into the consumer's operand load, the entry point exists for operator-level parity. */
int fd_gn_silu_apply(const void* x, const float* affine, void* out, int B, long long hw, int C, int dtype, void* stream);

/* One ResnetBlockBigGANpp.forward (layerspp.py:252-284): GroupNorm_0 + SiLU [+ FIR up/down of h and x] -> Conv_0 + time
 * bias -> GroupNorm_1 + SiLU -> Conv_1 [+ Conv_2(x) folded in] (+ x) -> / sqrt(2), as the launches the model uses
 * (2 convs, 2 finalize, optional FIR).  x0 / x1 = NHWC input (virtual concat), out = NHWC [B][H'][W'][cout].
 *   w0    = fd_conv_pack_weights(Conv_0.weight, NULL, cout, cin0, cin1, 3, 0, 0)  (up/down blocks: cin1 = 0)
 *   w1    = fd_conv_pack_weights(Conv_1.weight, Conv_2.weight or NULL, cout, cout, 0, 3, S0, S1) with (S0, S1) = the input
 *           split (cin0, cin1) when has_conv2 -- the shortcut conv runs as extra K steps on x (resampled for up/down)
 *   bias0 = [bias0_rows][cout] from fd_temb_bias (Conv_0.bias + Dense_0(silu(temb))),  bias1 = Conv_1.bias (+ Conv_2.bias) */
typedef struct fd_resblock_desc {
  int cin0, cin1, cout, up, down, has_conv2;
  const float *gn0_gamma, *gn0_beta, *gn1_gamma, *gn1_beta;
  const void* w0; const float* bias0; int bias0_rows;
  const void* w1; const float* bias1;
} fd_resblock_desc;
size_t fd_resblock_workspace_bytes(const fd_resblock_desc* d, int B, int H, int W, int dtype);
int fd_resblock(const fd_resblock_desc* d, const void* x0, const void* x1, void* out, int B, int H, int W, int dtype, void* ws,
                size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Front / back end
 * ---------------------------------------------------------------------------------------------- */

/* normalize_noisy('noisy') + torch.stft(n_fft, hop, sym-Hann, center/reflect, onesided) + amplitude
 * compression beta*|X|^alpha*e^{j angle X} + zero pad of the frame axis to T_pad
 * (util/other.py:55-82, feature_extractors.py:86-96,118-128, util/other.py:25-52).
 * y [B][L] f32 -> Y [B][1][n_fft/2+1][T_pad] complex64, normfac [B] f32.
 * ws: fd_stft_workspace_bytes(B, L, n_fft, hop) bytes of scratch. */
/* The DFT matrices / window envelope of one (n_fft, hop) live in a plan the CALLER owns (2 x 9.4 MB on the current device for
 * n_fft = 1534): fd_stft_plan_create allocates and uploads (synchronous, init time), the transforms themselves allocate
 * nothing and keep no state. */
typedef struct fd_stft_plan fd_stft_plan;
int fd_stft_plan_create(int n_fft, int hop, fd_stft_plan** out);
void fd_stft_plan_destroy(fd_stft_plan* plan);
size_t fd_stft_workspace_bytes(int B, int L, int n_fft, int hop);
/* normalize != 0: per-clip max-abs normalisation (normalize_mode 'noisy'); 0: normfac = 1 (normalize_mode 'none', util/other.py:70) */
int fd_stft_compress(const fd_stft_plan* plan, const float* y, int B, int L, float alpha, float beta, int normalize,
                     float* normfac, float* Y, int T_pad, void* ws, size_t ws_bytes, void* stream);
/* Inverse: slice [:T] -> X/beta -> |.|^(1/alpha) -> torch.istft(length=L) -> * normfac
 * (model.py:165-190, feature_extractors.py:98-109,130-139).  normfac may be NULL. */
int fd_decompress_istft(const fd_stft_plan* plan, const float* X, int B, int T, int T_pad, float alpha, float beta,
                        const float* normfac, float* y, int L, void* ws, size_t ws_bytes, void* stream);
/* Ragged batches (round 6).  The reference's driver enhances a directory FILE BY FILE, every file its own length (enhance.py:96-137,
 * model.py:129-163,476-528).  These variants take the files whose spectrograms pad to the same T_pad (util/other.py:25-52) as ONE
 * batch: y / the output are [B][L] rows with L = the longest clip, `lengths` (DEVICE int32 [B], n_fft/2 < lengths[b] <= L) the clips'
 * own sample counts.  Clip b gets exactly the arithmetic of a call with that clip alone: reflect padding at ITS end, its own
 * 1 + lengths[b]/hop frames (the rest zero, as pad_spec would leave them), torch.istft(length = lengths[b]) with its own overlap-add
 * envelope -- the result is bit-identical to the one-clip call; samples [lengths[b], L) of an output row are zero.  The caller
 * guarantees fd_padded_frames(fd_num_frames(lengths[b], hop)) == T_pad for every b; the kernels clamp a length into (n_fft/2, L]. */
int fd_stft_compress_ragged(const fd_stft_plan* plan, const float* y, const int* lengths, int B, int L, float alpha, float beta,
                            int normalize, float* normfac, float* Y, int T_pad, void* ws, size_t ws_bytes, void* stream);
/* T = 1 + L / hop (the frame count of the row length). */
int fd_decompress_istft_ragged(const fd_stft_plan* plan, const float* X, const int* lengths, int B, int T, int T_pad, float alpha,
                               float beta, const float* normfac, float* y, int L, void* ws, size_t ws_bytes, void* stream);
/* CompressAmplitudesAndScale.forward (inverse = 0: beta |x|^alpha e^{j angle x}) / .invert (inverse = 1) on n complex64
 * values (feature_extractors.py:118-139) as a stand-alone pass; X == Y allowed. */
int fd_compress_spec(const float* X, float* Y, long long n, float alpha, float beta, int inverse, void* stream);
int fd_num_frames(int L, int hop);     /* 1 + L / hop */
int fd_padded_frames(int T);           /* next multiple of 64 */

/* ------------------------------------------------------------------------------------------------
 * Whole-model entry points
 * ---------------------------------------------------------------------------------------------- */
typedef struct fd_model fd_model;

typedef struct fd_model_config {
  int nf;                /* 64 */
  int ch_mult[8];        /* {4,4,4,2} */
  int num_levels;        /* 4 */
  int num_res_blocks;    /* 1 */
  int n_fft;             /* 1534 */
  int hop;               /* 384 */
  float alpha, beta;     /* 0.3, 0.33 */
  int act_dtype;         /* FD_BF16 (bf16 storage + bf16 MFMA) [| FD_WINOGRAD | FD_WINOGRAD_LOWRES | FD_WINOGRAD_AUTO | FD_LOW_LATENCY], FD_F32 (f32 storage + exact f32 MFMA), FD_F32 | FD_BF16_OPERANDS or FD_F32 | FD_BF16X3_OPERANDS */
} fd_model_config;

int fd_model_create(const fd_model_config* cfg, fd_model** out);
void fd_model_destroy(fd_model* m);
/* Number of parameter tensors the model expects, and the i-th name / shape (reference state_dict
 * layout `backbone.all_modules.<i>.<...>`). */
int fd_model_num_params(const fd_model* m);
int fd_model_param_info(const fd_model* m, int i, const char** name, int* ndim, int shape[4]);
/* Copy one parameter (float32, HOST pointer, contiguous, reference shape) into the model. */
int fd_model_set_param(fd_model* m, const char* name, const float* host_data, long long numel);
/* sigma_y: per-frequency curve [n_freq] (float64 host) or a scalar (n = 1) (model.py:399-419). */
int fd_model_set_sigma_y(fd_model* m, const double* host_sigma, int n);
/* Pack weights for MFMA and upload; must be called after all fd_model_set_param and before forward. */
int fd_model_finalize(fd_model* m, void* stream);

size_t fd_model_workspace_bytes(const fd_model* m, int B, int T_pad);
/* v = NCSNpp(x, y, t): x, y, v complex64 [B][1][F][T_pad]; t float32 [nt] with nt in {1, B}
 * (FlowModel.forward, model.py:470-474; ncsnpp.py:254-399). */
int fd_ncsnpp_forward(fd_model* m, const float* x, const float* y, const float* t, int nt, float* v, int B, int T_pad,
                      void* ws, size_t ws_bytes, void* stream);
/* x0 = Y + sigma_fac * (sigma_y * noise) (model.py:512,530-536), then the fixed-step solve over
 * t_span = linspace(0,1,N+1) (:513-514) -- torchdyn fixed-step semantics restated (oracle/flowdec_oracle.py
 * odeint_fixed).  X (in: Y; out: final state) complex64 [B][1][F][T_pad]; noise complex64 same shape
 * (standard complex normal, supplied by the caller so results are reproducible); traj (optional)
 * receives all N+1 states.  use_graph != 0 captures the whole solve into a hipGraph cached per
 * (B, T_pad, N, solver) and replays it. */
int fd_ode_solve(fd_model* m, const float* Y, const float* noise, float sigma_fac, int N, int solver, float* X_out,
                 float* traj, int B, int T_pad, void* ws, size_t ws_bytes, int use_graph, void* stream);
/* Adaptive Dormand-Prince 5(4) over t_span = linspace(0, 1, N+1) (solver='dopri5' of the reference's torchdyn NeuralODE,
 * model.py:487-514; controller semantics restated, see oracle/).  Host-driven: the error ratio of every attempted step is
 * read back, so the call synchronises `stream` and cannot be graph-captured.  traj (optional) receives the N+1 states at
 * the t_span checkpoints, *nfe_out the number of vector-field evaluations. */
size_t fd_ode_adaptive_workspace_bytes(const fd_model* m, int B, int T_pad);
int fd_ode_solve_adaptive(fd_model* m, const float* Y, const float* noise, float sigma_fac, int N, float atol, float rtol,
                          float* X_out, float* traj, int* nfe_out, int B, int T_pad, void* ws, size_t ws_bytes, void* stream);
/* Same driver with the embedded pair chosen by id: Dormand-Prince 5(4) ('dopri5') or Tsitouras 5(4) ('tsit5', the default solver of
 * torchdyn's NeuralODE); both 7 stages with FSAL, the same controller. */
#define FD_ADAPTIVE_DOPRI5 0
#define FD_ADAPTIVE_TSIT5 1
int fd_ode_solve_adaptive_method(fd_model* m, const float* Y, const float* noise, float sigma_fac, int N, int method, float atol, float rtol,
                                 float* X_out, float* traj, int* nfe_out, int B, int T_pad, void* ws, size_t ws_bytes, void* stream);
size_t fd_enhance_workspace_bytes(const fd_model* m, int B, int L);
/* Byte offset, inside the workspace of fd_enhance / fd_score_enhance / fd_regression_enhance, of the B float32 normalisation
 * factors the front end computed (EnhancementModel._preprocess' `normfac`, model.py:156-162); valid after the call. */
size_t fd_enhance_normfac_offset(const fd_model* m, int B, int L);
/* FlowModel.enhance (model.py:476-528) end to end on device buffers: y [B][L] f32 -> x_hat [B][L] f32. */
int fd_enhance(fd_model* m, const float* y, const float* noise, float sigma_fac, int N, int solver, float* x_hat, int B,
               int L, void* ws, size_t ws_bytes, int use_graph, void* stream);
/* FlowModel.enhance on a ragged batch (see "Ragged batches" above): workspace as for fd_enhance(B, L); noise [B][1][F][T_pad];
 * clip b bit-identical to fd_enhance on that clip alone with the same noise.  A captured graph is keyed on the POINTER `lengths`
 * (its contents may change between replays). */
int fd_enhance_ragged(fd_model* m, const float* y, const int* lengths, const float* noise, float sigma_fac, int N, int solver,
                      float* x_hat, int B, int L, void* ws, size_t ws_bytes, int use_graph, void* stream);
/* normalize_mode of the model's front end: 1 = 'noisy' (default), 0 = 'none' (model.py:52, util/other.py:70). */
int fd_model_set_normalize(fd_model* m, int normalize);

/* ---- ScoreDec / regression baselines on the same backbone -------------------------------
 * ScoreModel.enhance (model.py:630-657) with the predictor-corrector sampler of sampling/__init__.py:32-72 on the OUVE
 * SDE (sdes.py:132-206): predictors reverse_diffusion / euler_maruyama / none (sampling/predictors.py:48-83), correctors
 * ald / none (sampling/correctors.py:42-80); the score is -backbone(x, y, t) / std(t) (model.py:613-628). */
#define FD_PREDICTOR_REVERSE_DIFFUSION 0
#define FD_PREDICTOR_EULER_MARUYAMA 1
#define FD_PREDICTOR_NONE 2
#define FD_CORRECTOR_ALD 0
#define FD_CORRECTOR_NONE 1
typedef struct fd_score_config {
  float theta, sigma_min, sigma_max; /* OUVESDE(theta, sigma_min, sigma_max), config/model/sde/ouve_final.yaml */
  float t_eps;                       /* ScoreModel.t_eps: timesteps = linspace(1, t_eps, N) */
  float snr;                         /* corrector target SNR */
  int N;                             /* reverse steps */
  int predictor, corrector, corrector_steps;
  int denoise;                       /* != 0: return the noise-free mean of the last predictor step */
} fd_score_config;
/* Number of Gaussian draws the sampler consumes: 1 (prior) + N * (corrector_steps [ald] + 1 [predictor != none]). */
int fd_score_num_draws(const fd_score_config* cfg);
/* y [B][L] f32 -> x_hat [B][L] f32.  noise = [fd_score_num_draws][B][n_freq][T_pad] complex64 standard normal, consumed in
 * the order of the reference's randn_like calls.  Workspace: fd_enhance_workspace_bytes(m, B, L). */
int fd_score_enhance(fd_model* m, const float* y, const float* noise, const fd_score_config* cfg, float* x_hat, int B, int L,
                     void* ws, size_t ws_bytes, int use_graph, void* stream);
/* One score-network evaluation in the form the black-box ODE sampler needs (sampling/__init__.py:75-146, which drives
 * scipy.integrate.solve_ivp on the host): x, Y, out = complex64 [B][1][n_freq][T_pad]; workspace fd_model_workspace_bytes.
 *   FD_SCORE_DRIFT_PF : out = theta (Y - x) - 0.5 g(t)^2 score(x, Y, t)     (probability-flow drift, sdes.py:93-109)
 *   FD_SCORE_DRIFT    : out = theta (Y - x) -     g(t)^2 score(x, Y, t)     (reverse-SDE drift)
 *   FD_SCORE_DENOISE  : out = mean of one reverse-diffusion predictor step at t (predictors.py:61-71), dt = 1 / cfg->N */
#define FD_SCORE_DRIFT_PF 0
#define FD_SCORE_DRIFT 1
#define FD_SCORE_DENOISE 2
int fd_score_eval(fd_model* m, const float* x, const float* Y, float t, const fd_score_config* cfg, int mode, float* out, int B,
                  int T_pad, void* ws, size_t ws_bytes, void* stream);
/* RegressionModel.enhance (model.py:566-578): x_hat = iSTFT(backbone(Y, Y, t = 0)). */
int fd_regression_enhance(fd_model* m, const float* y, float* x_hat, int B, int L, void* ws, size_t ws_bytes, int use_graph,
                          void* stream);

/* Per-launch timing of the dominant kernel (conv MFMA) measured with HIP events on the launch stream;
 * used by bench.py for the roofline object.  enable != 0 starts recording (forces eager launches). */
int fd_profile_enable(fd_model* m, int enable);
int fd_profile_read(fd_model* m, double* conv_ms_total, long long* conv_launches, double* conv_flops_total,
                    double* conv_bytes_total /* algorithmic HBM bytes: operands read once + output written once */);
/* The multiply-add FLOPs the convolution launches actually EXECUTED since fd_profile_enable / the last call of this function (reads and
 * resets its own counter; call it before or after fd_profile_read): a Winograd F(4,3) launch executes half, an F(2,3) launch two thirds
 * of the direct count of its 3x3 part.  conv_flops_total of fd_profile_read stays the DIRECT convolution's count. */
int fd_profile_read_executed(fd_model* m, double* conv_flops_executed);
/* Same for the HBM-bound FIR resampling launches (fd_fir_resample inside the model): total time, launches and algorithmic
 * bytes (input read once + every output written once) since fd_profile_enable. */
int fd_profile_read_fir(fd_model* m, double* ms_total, long long* launches, double* bytes_total);
/* Front / back end (ComplexSTFT + compression, feature_extractors.py:86-139): per-kernel time of the LAST forward and the last
 * inverse transform recorded while profiling was on, ms6 = {absmax + framing, forward DFT GEMM, compression, decompression,
 * inverse DFT GEMM, overlap-add}; calls2 = number of forward / inverse calls seen.  The plan-level pair does the same for a
 * caller-owned plan (fd_stft_compress / fd_decompress_istft); fd_profile_enable switches it on for the model's own plan. */
int fd_profile_read_stft(fd_model* m, double* ms6, int* calls2);
int fd_stft_plan_profile(fd_stft_plan* plan, int enable);
int fd_stft_plan_profile_read(fd_stft_plan* plan, double* ms6, int* calls2);

/* ------------------------------------------------------------------------------------------------
 * NDAC codec: the Descript-Audio-Codec architecture whose output FlowDec post-filters.  Reference call
 * sites: demo.ipynb cell 2 (`DAC.load(.../weights.pth)`), cell 3 (`preprocess`, `encode(x, n_quantizers=nq)`,
 * `quantizer.from_codes(codes)`, `decode(zq)`); arithmetic = descript-audio-codec==1.0.0 (requirements.txt:4), a third-party
 * package that is not under /root/reference: restated from the published algorithm, PARITY UNPINNED (oracle/ndac_oracle.py).
 * Layout [B][C][T] float32 like nn.Conv1d.  Weights are passed EFFECTIVE (weight norm g * v / ||v|| folded by the caller:
 * flowdec_amd/ndac.py), under their state_dict names with `.weight` in place of `.weight_g` / `.weight_v`.
 * ---------------------------------------------------------------------------------------------- */
/* Operator level (stateless, re-entrant): nn.Conv1d / nn.ConvTranspose1d with the DAC surroundings fused in.
 *   alpha_in [Ci] or NULL : Snake activation x + sin^2(alpha x) / (alpha + 1e-9) applied to the INPUT (zero padding after it)
 *   residual [B][Co][To] or NULL : added to the result (ResidualUnit: x + block(x));  tanh_out != 0 : tanh of the result
 * w: [Co][Ci][K] (conv) / [Ci][Co][K] (transposed); To = (T + 2 p - d (K - 1) - 1) / s + 1  resp.  (T - 1) s - 2 p + K. */
int fd_conv1d(const float* x, const float* w, const float* bias, const float* alpha_in, const float* residual, float* out, int B, int Ci,
              int T, int Co, int K, int stride, int padding, int dilation, int tanh_out, void* stream);
int fd_conv_transpose1d(const float* x, const float* w, const float* bias, const float* alpha_in, float* out, int B, int Ci, int T, int Co,
                        int K, int stride, int padding, void* stream);

typedef struct fd_ndac fd_ndac;
typedef struct fd_ndac_config {   /* dac.DAC.__init__ keyword arguments (the `metadata["kwargs"]` of a weights.pth) */
  int encoder_dim;                /* 64 */
  int encoder_rates[8];           /* e.g. {2,4,8,8}; hop length = their product */
  int n_encoder_rates;
  int latent_dim;                 /* encoder_dim * 2^n_encoder_rates unless the checkpoint says otherwise */
  int decoder_dim;                /* 1536 */
  int decoder_rates[8];           /* e.g. {8,8,4,2} */
  int n_decoder_rates;
  int n_codebooks, codebook_size, codebook_dim;   /* 9, 1024, 8 (codebook_dim <= 8) */
} fd_ndac_config;
int fd_ndac_create(const fd_ndac_config* cfg, fd_ndac** out);
void fd_ndac_destroy(fd_ndac* m);
int fd_ndac_hop_length(const fd_ndac* m);
int fd_ndac_num_params(const fd_ndac* m);
int fd_ndac_param_info(const fd_ndac* m, int i, const char** name, int* ndim, int shape[3]);
int fd_ndac_set_param(fd_ndac* m, const char* name, const float* host_data, long long numel);   /* float32 HOST pointer */
int fd_ndac_finalize(fd_ndac* m, void* stream);   /* uploads; normalises the codebooks (synchronous, init time) */
/* Arithmetic (bit flags).  FD_NDAC_EXACT: float32 on the vector ALUs in a defined operation order; the encoder's code indices are
 * bit-identical to oracle/ndac_oracle.py.  FD_NDAC_MFMA_DECODER (the default): the decoder's wide convolutions run on the matrix
 * cores with both operands split into two bf16 terms (three products, f32 accumulation: ~1e-6 relative per layer) and the hardware
 * sine in Snake; decode agrees with the exact path to ~1e-5 of the waveform peak.  FD_NDAC_MFMA_ENCODER (opt-in): the same for
 * the encoder; the latent agrees to ~1e-5, so a code index can differ from the exact path where two codebook entries are within
 * that distance of a tie (tests/test_hip_ndac.py bounds the fraction). */
#define FD_NDAC_EXACT 0
#define FD_NDAC_MFMA_DECODER 1
#define FD_NDAC_MFMA_ENCODER 2
int fd_ndac_set_precision(fd_ndac* m, int flags);
int fd_ndac_get_precision(const fd_ndac* m);
int fd_ndac_latent_frames(const fd_ndac* m, int L);    /* frames for L samples (L % hop == 0: L / hop) */
int fd_ndac_decoded_length(const fd_ndac* m, int T);   /* samples the decoder produces for T frames */
size_t fd_ndac_workspace_bytes(const fd_ndac* m, int B, int L);   /* covers encode of [B][L] and decode of its frames */
/* dac.DAC.encode (eval): x [B][L] (L % hop == 0: DAC.preprocess pads) -> z_q [B][latent][T] f32, codes [B][nq][T] int32,
 * latents [B][nq * codebook_dim][T] or NULL.  n_quantizers <= 0 or > n_codebooks: all of them. */
int fd_ndac_encode(fd_ndac* m, const float* x, int B, int L, int n_quantizers, float* z_q, int* codes, float* latents, void* ws,
                   size_t ws_bytes, void* stream);
/* ResidualVectorQuantize.forward on a given latent z [B][latent][T] (ws: B * latent * T floats) / .from_codes */
int fd_rvq_encode(fd_ndac* m, const float* z, int B, int T, int n_quantizers, float* z_q, int* codes, float* latents, void* ws,
                  size_t ws_bytes, void* stream);
int fd_rvq_from_codes(fd_ndac* m, const int* codes, int B, int n_quantizers, int T, float* z_q, void* stream);
/* dac.DAC.decode: z [B][latent][T] -> audio [B][fd_ndac_decoded_length(T)] in (-1, 1) */
int fd_ndac_decode(fd_ndac* m, const float* z, int B, int T, float* audio, void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FLOWDEC_HIP_H */
