"""ctypes binding of libflowdec_hip.so (the C ABI in include/flowdec_hip.h).

The HIP library is the ONLY compute path of this package: if it is missing, importing the ops
raises -- there is no CPU / eager-PyTorch fallback.  PyTorch is used for device memory, streams
and torch.distributed only.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FLOWDEC_HIP_LIB") or os.path.join(_HERE, "libflowdec_hip.so")  # (env: kernel-variant A/B runs)

FD_F32, FD_BF16 = 0, 1
FD_WINOGRAD = 0x100  # algorithm flag OR-ed into a dtype argument (include/flowdec_hip.h)
FD_WINOGRAD4 = 0x80000  # Winograd F(4,3) along W, 256-cout workgroups (conv_wino4.hip)
FD_TILE_REVERSED = 0x100000  # with FD_WINOGRAD4: tiles in descending order (same bits)
FD_WINOGRAD44 = 0x200000  # with FD_F32: 2-D Winograd F(4x4, 3x3) in exact float32 (conv_wino44f.hip)
FD_WINOGRAD_LOWRES = 0x200
FD_WINOGRAD_AUTO = 0x400
FD_LOW_LATENCY = 0x800
FD_BF16_OPERANDS = 0x10000  # with FD_F32: f32 storage, bf16 MFMA operands (precision='mixed')
FD_NO_SIDE_STREAM = 0x40000  # fd_model_config.act_dtype: side branches stay on the caller's stream
FD_EBUSY = -5
FD_BF16X3_OPERANDS = 0x20000  # with FD_F32: operands as hi + lo bf16 pairs, three bf16 MFMAs per product (precision='bf16x3')
FD_TILE = {0: 0, 32: 0x1000, 64: 0x2000, 128: 0x3000, "64c": 0x4000, "32c": 0x5000, "duo": 0x6000, "persist": 0x7000}  # fd_conv2d: output channels per workgroup (0 = default)
SOLVERS = {"euler": 0, "midpoint": 1, "heun2": 2, "heun2_eulerlast": 3}
ADAPTIVE_SOLVERS = {"dopri5": 0, "tsit5": 1}   # FD_ADAPTIVE_*

c_void_p, c_int, c_float, c_ll, c_size_t = C.c_void_p, C.c_int, C.c_float, C.c_longlong, C.c_size_t


class FdModelConfig(C.Structure):
    _fields_ = [("nf", c_int), ("ch_mult", c_int * 8), ("num_levels", c_int), ("num_res_blocks", c_int),
                ("n_fft", c_int), ("hop", c_int), ("alpha", c_float), ("beta", c_float), ("act_dtype", c_int)]


class FdNdacConfig(C.Structure):
    _fields_ = [("encoder_dim", c_int), ("encoder_rates", c_int * 8), ("n_encoder_rates", c_int), ("latent_dim", c_int), ("decoder_dim", c_int),
                ("decoder_rates", c_int * 8), ("n_decoder_rates", c_int), ("n_codebooks", c_int), ("codebook_size", c_int), ("codebook_dim", c_int)]


class FdResblockDesc(C.Structure):
    _fields_ = [("cin0", c_int), ("cin1", c_int), ("cout", c_int), ("up", c_int), ("down", c_int), ("has_conv2", c_int),
                ("gn0_gamma", c_void_p), ("gn0_beta", c_void_p), ("gn1_gamma", c_void_p), ("gn1_beta", c_void_p),
                ("w0", c_void_p), ("bias0", c_void_p), ("bias0_rows", c_int), ("w1", c_void_p), ("bias1", c_void_p)]


class FdScoreConfig(C.Structure):
    _fields_ = [("theta", c_float), ("sigma_min", c_float), ("sigma_max", c_float), ("t_eps", c_float), ("snr", c_float),
                ("N", c_int), ("predictor", c_int), ("corrector", c_int), ("corrector_steps", c_int), ("denoise", c_int)]


PREDICTORS = {"reverse_diffusion": 0, "euler_maruyama": 1, "none": 2}
CORRECTORS = {"ald": 0, "none": 1}

# name -> (restype, [argtypes]); must list every function declared in include/flowdec_hip.h
_P = c_void_p
SIGNATURES = {
    "fd_last_error": (C.c_char_p, []),
    "fd_version": (c_int, []),
    "fd_device_info": (c_int, [C.POINTER(c_int)]),
    "fd_upfirdn2d": (c_int, [_P, _P, _P] + [c_int] * 15 + [_P]),
    "fd_upfirdn2d_out_size": (c_int, [c_int] * 6),
    "fd_fused_bias_act": (c_int, [_P, _P, _P, c_ll, c_int, c_int, c_int, c_float, c_float, _P]),
    "fd_fir_resample": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "fd_conv_in": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "fd_channel_sums_tiles": (c_int, [c_int, c_int]),
    "fd_channel_sums": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "fd_gn_silu_apply": (c_int, [_P, _P, _P, c_int, c_ll, c_int, c_int, _P]),
    "fd_resblock_workspace_bytes": (c_size_t, [C.POINTER(FdResblockDesc), c_int, c_int, c_int, c_int]),
    "fd_resblock": (c_int, [C.POINTER(FdResblockDesc), _P, _P, _P, c_int, c_int, c_int, c_int, _P, c_size_t, _P]),
    "fd_gn_finalize": (c_int, [_P, c_int, c_int, c_int, _P, c_int, c_int, c_int, _P, _P, _P, c_int, c_int, c_ll, c_float, _P]),
    "fd_conv_packed_bytes": (c_ll, [c_int] * 7),
    "fd_conv_pack_weights": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "fd_conv_cout_pad": (c_int, [c_int]),
    "fd_conv_stats_tiles": (c_int, [c_int, c_int]),
    "fd_conv2d": (c_int, [_P, c_int, _P, c_int, _P, _P, c_int, _P, c_int, _P, _P, c_int, _P, c_float, _P, c_int, _P,
                          c_int, c_int, c_int, c_int, c_int, _P]),
    "fd_time_embedding": (c_int, [_P, c_int, _P, c_int, _P, _P, _P, _P, _P, _P]),
    "fd_temb_bias": (c_int, [_P, c_int, c_int, _P, _P, _P, c_int, _P, _P]),
    "fd_stft_workspace_bytes": (c_size_t, [c_int] * 4),
    "fd_stft_plan_create": (c_int, [c_int, c_int, C.POINTER(c_void_p)]),
    "fd_stft_plan_destroy": (None, [_P]),
    "fd_stft_compress": (c_int, [_P, _P, c_int, c_int, c_float, c_float, c_int, _P, _P, c_int, _P, c_size_t, _P]),
    "fd_decompress_istft": (c_int, [_P, _P, c_int, c_int, c_int, c_float, c_float, _P, _P, c_int, _P, c_size_t, _P]),
    "fd_stft_compress_ragged": (c_int, [_P, _P, _P, c_int, c_int, c_float, c_float, c_int, _P, _P, c_int, _P, c_size_t, _P]),
    "fd_decompress_istft_ragged": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_float, c_float, _P, _P, c_int, _P, c_size_t, _P]),
    "fd_compress_spec": (c_int, [_P, _P, c_ll, c_float, c_float, c_int, _P]),
    "fd_num_frames": (c_int, [c_int, c_int]),
    "fd_padded_frames": (c_int, [c_int]),
    "fd_model_create": (c_int, [C.POINTER(FdModelConfig), C.POINTER(_P)]),
    "fd_model_destroy": (None, [_P]),
    "fd_model_num_params": (c_int, [_P]),
    "fd_model_param_info": (c_int, [_P, c_int, C.POINTER(C.c_char_p), C.POINTER(c_int), C.POINTER(c_int * 4)]),
    "fd_model_set_param": (c_int, [_P, C.c_char_p, _P, c_ll]),
    "fd_model_set_sigma_y": (c_int, [_P, _P, c_int]),
    "fd_model_finalize": (c_int, [_P, _P]),
    "fd_model_workspace_bytes": (c_size_t, [_P, c_int, c_int]),
    "fd_ncsnpp_forward": (c_int, [_P, _P, _P, _P, c_int, _P, c_int, c_int, _P, c_size_t, _P]),
    "fd_ode_solve": (c_int, [_P, _P, _P, c_float, c_int, c_int, _P, _P, c_int, c_int, _P, c_size_t, c_int, _P]),
    "fd_ode_adaptive_workspace_bytes": (c_size_t, [_P, c_int, c_int]),
    "fd_ode_solve_adaptive": (c_int, [_P, _P, _P, c_float, c_int, c_float, c_float, _P, _P, C.POINTER(c_int), c_int, c_int, _P, c_size_t, _P]),
    "fd_ode_solve_adaptive_method": (c_int, [_P, _P, _P, c_float, c_int, c_int, c_float, c_float, _P, _P, C.POINTER(c_int), c_int, c_int, _P, c_size_t, _P]),
    "fd_enhance_workspace_bytes": (c_size_t, [_P, c_int, c_int]),
    "fd_enhance_normfac_offset": (c_size_t, [_P, c_int, c_int]),
    "fd_model_set_normalize": (c_int, [_P, c_int]),
    "fd_calibrate_mfma": (c_int, [_P, c_int, c_int, _P, _P, _P]),
    "fd_enhance": (c_int, [_P, _P, _P, c_float, c_int, c_int, _P, c_int, c_int, _P, c_size_t, c_int, _P]),
    "fd_enhance_ragged": (c_int, [_P, _P, _P, _P, c_float, c_int, c_int, _P, c_int, c_int, _P, c_size_t, c_int, _P]),
    "fd_score_num_draws": (c_int, [C.POINTER(FdScoreConfig)]),
    "fd_score_enhance": (c_int, [_P, _P, _P, C.POINTER(FdScoreConfig), _P, c_int, c_int, _P, c_size_t, c_int, _P]),
    "fd_score_eval": (c_int, [_P, _P, _P, c_float, C.POINTER(FdScoreConfig), c_int, _P, c_int, c_int, _P, c_size_t, _P]),
    "fd_regression_enhance": (c_int, [_P, _P, _P, c_int, c_int, _P, c_size_t, c_int, _P]),
    "fd_profile_enable": (c_int, [_P, c_int]),
    "fd_profile_read_executed": (c_int, [_P, C.POINTER(C.c_double)]),
    "fd_profile_read": (c_int, [_P, C.POINTER(C.c_double), C.POINTER(c_ll), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "fd_profile_read_fir": (c_int, [_P, C.POINTER(C.c_double), C.POINTER(c_ll), C.POINTER(C.c_double)]),
    "fd_profile_read_stft": (c_int, [_P, C.POINTER(C.c_double * 6), C.POINTER(c_int * 2)]),
    "fd_conv1d": (c_int, [_P, _P, _P, _P, _P, _P] + [c_int] * 9 + [_P]),
    "fd_conv_transpose1d": (c_int, [_P, _P, _P, _P, _P] + [c_int] * 7 + [_P]),
    "fd_ndac_create": (c_int, [C.POINTER(FdNdacConfig), C.POINTER(_P)]),
    "fd_ndac_destroy": (None, [_P]),
    "fd_ndac_hop_length": (c_int, [_P]),
    "fd_ndac_num_params": (c_int, [_P]),
    "fd_ndac_param_info": (c_int, [_P, c_int, C.POINTER(C.c_char_p), C.POINTER(c_int), C.POINTER(c_int * 3)]),
    "fd_ndac_set_param": (c_int, [_P, C.c_char_p, _P, c_ll]),
    "fd_ndac_finalize": (c_int, [_P, _P]),
    "fd_ndac_set_precision": (c_int, [_P, c_int]),
    "fd_ndac_get_precision": (c_int, [_P]),
    "fd_ndac_latent_frames": (c_int, [_P, c_int]),
    "fd_ndac_decoded_length": (c_int, [_P, c_int]),
    "fd_ndac_workspace_bytes": (c_size_t, [_P, c_int, c_int]),
    "fd_ndac_encode": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P, c_size_t, _P]),
    "fd_rvq_encode": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P, c_size_t, _P]),
    "fd_rvq_from_codes": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P]),
    "fd_ndac_decode": (c_int, [_P, _P, c_int, c_int, _P, _P, c_size_t, _P]),
    "fd_stft_plan_profile": (c_int, [_P, c_int]),
    "fd_stft_plan_profile_read": (c_int, [_P, C.POINTER(C.c_double * 6), C.POINTER(c_int * 2)]),
}

_lib = None


def load():
    """Load libflowdec_hip.so; raises if it has not been built (`python -m flowdec_amd.build`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is the only compute path of flowdec_amd. "
            "Build it with `python -m flowdec_amd.build` (needs hipcc, targets gfx950).")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    """Non-zero return -> RuntimeError, like the reference's TORCH_CHECK (op/upfirdn2d.cpp:34-42)."""
    if rc != 0:
        msg = load().fd_last_error()
        raise RuntimeError("flowdec_hip error %d: %s" % (rc, msg.decode() if msg else "?"))


def ptr(t):
    """Device (or host) pointer of a tensor, None -> NULL."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def stream():
    """Raw hipStream_t of torch's current stream (the launch stream of every op)."""
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dtype_id(dt):
    if dt == torch.float32:
        return FD_F32
    if dt == torch.bfloat16:
        return FD_BF16
    raise RuntimeError(f"flowdec_amd: unsupported activation dtype {dt} (float32 / bfloat16 only)")


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("flowdec_amd: tensor must live on the GPU (HIP is the only compute path)")
