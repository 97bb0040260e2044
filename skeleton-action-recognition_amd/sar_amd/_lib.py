"""ctypes binding of libsar_hip.so (the C ABI declared in include/sar_hip.h).

The product path has NO fallback: if the HIP library is missing or a call fails,
an exception is raised.  Nothing here imports the CPU oracle.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SAR_HIP_LIB") or os.path.join(_HERE, "libsar_hip.so")   # env override: diagnostic builds

SAR_CONV_GRAPH, SAR_CONV_TEMPORAL = 0, 1
SAR_EPI_NONE, SAR_EPI_STATS, SAR_EPI_MASK, SAR_EPI_ADD, SAR_EPI_ADD_GATE = 0, 1, 2, 3, 4
SAR_E_ARG, SAR_E_UNSUP = -1, -2
SAR_GRAPH_WT_BF16_EXACT = 1
SAR_GRAPH_FEW_DENSE = 4
SAR_GRAPH_SLICE0_IDENTITY = 8
SAR_GRAPH_FEW_DENSE_SHIFT = 8
SAR_GRAPH_ONE_TILE_WG = 16
SAR_GRAPH_AUX_EVEN_FRAMES = 32
SAR_C2D_AUX_EVEN_PIXELS = 1
SAR_SPLIT = {"bf16x1": 1, "bf16x3": 3, "bf16x6": 6, "bf16x9": 9, "f16x3": 103, "f16x3s": 104, "f16x3a": 105}   # include/sar_hip.h SAR_SPLIT_*

_fp = C.c_void_p  # every device pointer crosses the ABI as void*


class ConvDesc(C.Structure):
    _fields_ = [
        ("mode", C.c_int32), ("transposed", C.c_int32), ("B", C.c_int32), ("V", C.c_int32),
        ("T_src", C.c_int32), ("T_out", C.c_int32), ("Kc", C.c_int32), ("M", C.c_int32),
        ("taps", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("pro_relu", C.c_int32),
        ("epi", C.c_int32), ("nz", C.c_int32 * 3), ("g_flags", C.c_int32), ("reserved0", C.c_int32),
        ("src", _fp), ("ld_src", C.c_int64), ("out", _fp), ("ld_out", C.c_int64),
        ("W", _fp), ("w_stride_tap", C.c_int64), ("w_stride_c", C.c_int64), ("bias", _fp),
        ("pro_scale", _fp), ("pro_shift", _fp),
        ("g_idx", _fp), ("g_wt", _fp), ("g_colsum", _fp),
        ("aux", _fp), ("ld_aux", C.c_int64), ("aux_scale", _fp), ("aux_shift", _fp), ("aux_mean", _fp),
        ("partials", _fp), ("aux2", _fp), ("ld_aux2", C.c_int64), ("aux_mask", _fp),
    ]


class WgradDesc(C.Structure):
    _fields_ = [
        ("mode", C.c_int32), ("B", C.c_int32), ("V", C.c_int32), ("T_src", C.c_int32), ("T_out", C.c_int32),
        ("Kc", C.c_int32), ("M", C.c_int32), ("taps", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("pro_relu", C.c_int32), ("nz", C.c_int32 * 3), ("nsplit", C.c_int32), ("g_flags", C.c_int32),
        ("src", _fp), ("ld_src", C.c_int64), ("dout", _fp), ("ld_dout", C.c_int64),
        ("pro_scale", _fp), ("pro_shift", _fp),
        ("g_idx", _fp), ("g_wt", _fp), ("g_colsum", _fp),
        ("w_stride_tap", C.c_int64), ("w_stride_c", C.c_int64), ("wsize", C.c_int64), ("bsize", C.c_int64),
        ("slab", _fp),
    ]


class Conv2dDesc(C.Structure):
    _fields_ = [
        ("transposed", C.c_int32), ("B", C.c_int32), ("Kc", C.c_int32), ("M", C.c_int32),
        ("H_src", C.c_int32), ("W_src", C.c_int32), ("H_out", C.c_int32), ("W_out", C.c_int32),
        ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("pro_relu", C.c_int32), ("epi", C.c_int32), ("nsplit", C.c_int32), ("flags", C.c_int32),
        ("src", _fp), ("ld_src", C.c_int64), ("out", _fp), ("ld_out", C.c_int64),
        ("dout", _fp), ("ld_dout", C.c_int64),
        ("W", _fp), ("w_stride_tap", C.c_int64), ("w_stride_c", C.c_int64),
        ("pro_scale", _fp), ("pro_shift", _fp),
        ("aux", _fp), ("ld_aux", C.c_int64), ("aux_scale", _fp), ("aux_shift", _fp), ("aux_mean", _fp),
        ("partials", _fp), ("slab", _fp), ("ctx", _fp),
    ]


# name -> (restype, argtypes); every name must also be declared in include/sar_hip.h
_i, _i64, _f, _d = C.c_int, C.c_int64, C.c_float, C.c_double


class BnTail(C.Structure):
    """sar_bn_tail (include/sar_hip.h): the BatchNorm-backward finalisation folded into the reduce kernel's last workgroup"""
    _fields_ = [("ticket", _fp), ("count", C.c_double), ("gamma", _fp), ("rstd", _fp), ("dgamma", _fp), ("dbeta", _fp),
                ("k1", _fp), ("k2", _fp), ("k3", _fp), ("rgamma", _fp), ("rrstd", _fp), ("rdgamma", _fp), ("rdbeta", _fp),
                ("rk1", _fp), ("rk2", _fp), ("rk3", _fp)]
SIGNATURES = {
    "sar_version": (_i, []),
    "sar_last_error_string": (C.c_char_p, []),
    "sar_context_create": (_i, [C.POINTER(C.c_void_p)]),
    "sar_context_destroy": (_i, [_fp]),
    "sar_stream_create_cu_mask": (_i, [_fp, _i, C.POINTER(C.c_void_p)]),
    "sar_stream_destroy": (_i, [_fp]),
    "sar_struct_size": (_i, [_i]),
    "sar_conv_gemm_nparts": (_i, [C.POINTER(ConvDesc)]),
    "sar_conv_gemm_f32": (_i, [C.POINTER(ConvDesc), _fp]),
    "sar_conv_gemm_bf16_workspace_bytes": (_i64, [C.POINTER(ConvDesc)]),
    "sar_conv_gemm_bf16": (_i, [C.POINTER(ConvDesc), _fp, _fp]),
    "sar_pack_weights_bf16_batch": (_i, [_fp, _fp, _i, _i64, _fp, _fp]),
    "sar_conv_gemm_split_workspace_bytes": (_i64, [C.POINTER(ConvDesc), _i]),
    "sar_conv_gemm_split_nparts": (_i, [C.POINTER(ConvDesc)]),
    "sar_pack_weights_split_batch": (_i, [_fp, _fp, _i, _i64, _i, _fp, _fp, _fp]),
    "sar_conv_gemm_split": (_i, [C.POINTER(ConvDesc), _i, _fp, _fp, _fp, _fp]),
    "sar_conv_wgrad_split_blocks": (_i, [C.POINTER(WgradDesc), _i, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sar_conv_wgrad_split": (_i, [C.POINTER(WgradDesc), _i, _fp, _fp, _fp]),
    "sar_amax_f32": (_i, [_fp, _i, _i64, _i64, _fp, _fp]),
    "sar_bn_bound_f32": (_i, [_fp, _fp, _i, _d, _fp, _fp]),
    "sar_affine_bound_f32": (_i, [_fp, _fp, _i, _fp, _fp, _fp]),
    "sar_conv_wgrad_f32": (_i, [C.POINTER(WgradDesc), _fp]),
    "sar_conv_wgrad_bf16": (_i, [C.POINTER(WgradDesc), _fp]),
    "sar_slab_reduce_f32": (_i, [_fp, _i, _i64, _i64, _fp, _fp]),
    "sar_slab_reduce_batch_f32": (_i, [_fp, _i, _i64, _fp]),
    "sar_bn_finalize_f32": (_i, [_fp, _i, _i, _d, _f, _f, _i, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp]),
    "sar_bn_eval_affine_f32": (_i, [_fp, _fp, _fp, _fp, _f, _i, _fp, _fp, _fp]),
    "sar_bn_bwd_finalize_f32": (_i, [_fp, _i, _i64, _i64, _i, _i, _i, _i, _d, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp]),
    "sar_data_bn_stats_f32": (_i, [_fp, _i, _i, _i, _i, _i, _fp, _i, _fp, _fp]),
    "sar_data_bn_apply_f32": (_i, [_fp, _i, _i, _i, _i, _i, _fp, _i, _fp, _fp, _fp, _i64, _fp]),
    "sar_data_bn_bwd_reduce_f32": (_i, [_fp, _i, _i, _i, _i, _i, _fp, _i, _fp, _i64, _fp, _fp, _fp]),
    "sar_bn_add_relu_fwd_f32": (_i, [_fp, _fp, _fp, _i, _fp, _fp, _fp, _fp, _i, _i64, _i64, _fp]),
    "sar_bn_add_relu_bwd_reduce_f32": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i64, _i64, _fp]),
    "sar_bn_add_relu_fwd_mask_f32": (_i, [_fp, _fp, _fp, _i, _fp, _fp, _fp, _fp, _fp, _i, _i64, _i64, _fp]),
    "sar_bn_add_relu_bwd_reduce_mask_f32": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i64, _i64, _fp]),
    "sar_bn_add_relu_bwd_apply_mask_f32": (_i, [_fp] * 13 + [_i, _i64, _i64, _fp]),
    "sar_bn_add_relu_bwd_reduce_tail_f32": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i64, _i64, C.POINTER(BnTail), _fp]),
    "sar_bn_add_relu_bwd_apply_f32": (_i, [_fp] * 13 + [_i, _i64, _i64, _fp]),
    "sar_affine2_f32": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _i, _i64, _i64, _fp]),
    "sar_affine2_amax_f32": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i64, _i64, _fp]),
    "sar_bn_add_relu_fwd_mask_amax_f32": (_i, [_fp, _fp, _fp, _i, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i64, _i64, _fp]),
    "sar_bn_add_relu_bwd_apply_mask_amax_f32": (_i, [_fp] * 15 + [_i, _i64, _i64, _fp]),
    "sar_pool_fwd_f32": (_i, [_fp, _i64, _i, _i, _i, _i, _fp, _fp]),
    "sar_fc_fwd_f32": (_i, [_fp, _fp, _fp, _i, _i, _i, _fp, _fp]),
    "sar_softmax_ce_f32": (_i, [_fp, _fp, _i, _i, _f, _fp, _fp, _fp, _fp]),
    "sar_fc_bwd_f32": (_i, [_fp, _fp, _fp, _i, _i, _i, _fp, _fp, _fp, _fp]),
    "sar_pool_bwd_f32": (_i, [_fp, _i64, _i, _i, _i, _i, _fp, _fp]),
    "sar_sgd_nesterov_f32": (_i, [_fp, _fp, _fp, _i64, _fp, _f, _fp]),
    "sar_transpose_f32": (_i, [_fp, _fp, _i, _i, _i, _fp]),
    "sar_conv2d_nparts": (_i, [C.POINTER(Conv2dDesc)]),
    "sar_conv2d_gemm_f32": (_i, [C.POINTER(Conv2dDesc), _fp]),
    "sar_conv2d_wgrad_f32": (_i, [C.POINTER(Conv2dDesc), _fp]),
    "sar_conv2d_gemm_split_workspace_bytes": (_i64, [C.POINTER(Conv2dDesc), _i]),
    "sar_conv2d_gemm_split_nparts": (_i, [C.POINTER(Conv2dDesc)]),
    "sar_conv2d_gemm_split_slab_bytes": (_i64, [C.POINTER(Conv2dDesc)]),
    "sar_conv2d_gemm_split": (_i, [C.POINTER(Conv2dDesc), _i, _fp, _fp, _fp, _fp]),
    "sar_conv2d_wgrad_split_blocks": (_i, [C.POINTER(Conv2dDesc), _i, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sar_conv2d_wgrad_split": (_i, [C.POINTER(Conv2dDesc), _i, _fp, _fp, _fp]),
    "sar_permute3_f32": (_i, [_fp, _fp, _i, _i, _i, _i64, _i64, _i64, _fp]),
    "sar_permute3_batch_f32": (_i, [_fp, _fp, _fp, _i, _i64, _fp]),
    "sar_bn_relu_maxpool_fwd_f32": (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _i64, _i64, _fp]),
    "sar_bn_relu_maxpool_bwd_nparts": (_i, [_i, _i, _i]),
    "sar_bn_relu_maxpool_bwd_f32": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i64, _i64, _fp]),
    "sar_adam_f32": (_i, [_fp, _fp, _fp, _fp, _i64, _fp, _fp, _f, _f, _f, _fp]),
    "sar_vr_signal_f32": (_i, [_fp, _i, _i, _i, _i, _fp, _fp, _i, _fp, _fp, _fp, _fp, _fp]),
    "sar_stft_logmag_f32": (_i, [_fp, _fp, _i, _i, _i, _i, _fp, _i, _fp, _fp]),
    "sar_stft_logmag_bwd_workspace_floats": (_i64, [_i, _i, _i, _i]),
    "sar_stft_kernels_fwd_f32": (_i, [_fp, _fp, _i, _i, _i, _i, _fp, _fp, _i, _fp, _fp]),
    "sar_stft_kernels_bwd_workspace_floats": (_i64, [_i, _i, _i, _i, _i]),
    "sar_stft_kernels_bwd_f32": (_i, [_fp, _fp, _i, _i, _i, _i, _fp, _fp, _fp, _fp, _i, _fp, _fp, _i, _fp, _fp, _fp, _fp]),
    "sar_stft_logmag_bwd_f32": (_i, [_fp, _fp, _i, _i, _i, _i, _fp, _i, _fp, _fp, _fp, _fp, _fp]),
    "sar_vr_signal_bwd_nparts": (_i, [_i, _i]),
    "sar_vr_signal_bwd_f32": (_i, [_fp, _i, _i, _i, _i, _fp, _fp, _i, _fp, _fp, _fp, _fp, _fp, _fp]),
    "sar_upsample_workspace_bytes": (_i64, [_i, _i, _i, _i]),
    "sar_upsample_coef_doubles": (_i64, [_i, _i, _i, _i]),
    "sar_upsample_prepare_f64": (_i, [_fp, _i, _i, _i, _i, _fp, _i, _fp, _fp, _fp]),
    "sar_vr_signal_upsampled_f32": (_i, [_fp, _i, _i, _i, _i, _i, _fp, _fp, _i, _fp, _fp, _fp, _fp, _fp]),
    "sar_vr_signal_upsampled_bwd_f32": (_i, [_fp, _i, _i, _i, _i, _i, _fp, _fp, _i, _fp, _fp, _fp, _fp, _fp, _fp]),
    "sar_conv2d_stem_dgrad_f32": (_i, [_fp, _i64, _fp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _fp, _fp]),
    # dense (trainable) adjacency
    "sar_gin_nparts": (_i, [_i64]),
    "sar_gin_adjacency_f32": (_i, [_fp, _i, _i, _fp, _fp, _fp, _i, _fp, _fp]),
    "sar_graph_gather_sum_f32": (_i, [_fp, _i64, _fp, _fp, _fp, _fp, _i, _i, _i, _i64, _fp, _i64, _fp, _i64, _fp]),
    "sar_graph_gather_expand_f32": (_i, [_fp, _i64, _fp, _fp, _fp, _i, _i, _i, _i64, _fp, _i64, _fp]),
    "sar_gin_sum_fwd_f32": (_i, [_fp, _i64, _fp, _fp, _i, _i, _i64, _fp, _i64, _fp, _fp]),
    "sar_gin_bwd_reduce_f32": (_i, [_fp, _i64, _fp, _i64, _fp, _fp, _fp, _i, _i, _i64, _fp, _fp]),
    "sar_gin_bwd_apply_f32": (_i, [_fp, _i64, _fp, _i64, _fp, _fp, _fp, _fp, _fp, _i, _i, _i64, _fp, _i64, _fp]),
    "sar_gin_eps_grad_f32": (_i, [_fp, _fp, _i64, _fp, _fp, _fp]),
    "sar_graph_dense_nparts": (_i, [_i64]),
    "sar_graph_dense_fwd_f32": (_i, [_fp, _i64, _fp, _fp, _i64, _i, _i, _i, _i64, _fp, _fp, _i64, _fp]),
    "sar_graph_dense_bwd_data_f32": (_i, [_fp, _i64, _fp, _fp, _i64, _i, _i, _i, _i64, _fp]),
    "sar_graph_dense_dadj_slab_floats": (_i64, [_i, _i, _i, _i]),
    "sar_graph_dense_dadj_f32": (_i, [_fp, _i64, _fp, _i64, _i, _i, _i, _i64, _i, _fp, _fp, _fp]),
    # bf16 configuration: CN8 activations
    "sar_conv_gemm_cn8_nparts": (_i, [C.POINTER(ConvDesc)]),
    "sar_conv_gemm_cn8": (_i, [C.POINTER(ConvDesc), _fp, _fp]),
    "sar_conv_wgrad_cn8_tile_frames": (_i, [_i]),
    "sar_conv_wgrad_cn8": (_i, [C.POINTER(WgradDesc), _i, _fp]),
    "sar_bn_add_relu_fwd_cn8": (_i, [_fp, _fp, _fp, _i, _fp, _fp, _fp, _fp, _i, _i64, _i64, _fp]),
    "sar_bn_add_relu_bwd_reduce_cn8": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i64, _i64, _fp]),
    "sar_bn_add_relu_fwd_mask_cn8": (_i, [_fp, _fp, _fp, _i, _fp, _fp, _fp, _fp, _fp, _i, _i64, _i64, _fp]),
    "sar_bn_add_relu_bwd_reduce_mask_cn8": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i64, _i64, _fp]),
    "sar_bn_add_relu_bwd_apply_mask_cn8": (_i, [_fp] * 13 + [_i, _i64, _i64, _fp]),
    "sar_bn_add_relu_bwd_reduce_tail_cn8": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i64, _i64, C.POINTER(BnTail), _fp]),
    "sar_bn_add_relu_bwd_apply_cn8": (_i, [_fp] * 13 + [_i, _i64, _i64, _fp]),
    "sar_affine2_cn8": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _i, _i64, _i64, _fp]),
    "sar_data_bn_apply_cn8": (_i, [_fp, _i, _i, _i, _i, _i, _fp, _i, _fp, _fp, _fp, _i64, _fp]),
    "sar_data_bn_bwd_reduce_cn8": (_i, [_fp, _i, _i, _i, _i, _i, _fp, _i, _fp, _i64, _fp, _fp, _fp]),
    "sar_pool_fwd_cn8": (_i, [_fp, _i64, _i, _i, _i, _i, _fp, _fp]),
    "sar_pool_bwd_cn8": (_i, [_fp, _i64, _i, _i, _i, _i, _fp, _fp]),
    "sar_cn_to_cn8": (_i, [_fp, _i64, _fp, _i64, _i, _i64, _fp]),
    "sar_cn8_to_cn": (_i, [_fp, _i64, _fp, _i64, _i, _i64, _fp]),
    # box calibration (bench.py "box"; csrc/box_probe.hip)
    "sar_box_mfma": (_i, [_i, _i, _i, _fp, _fp, _fp]),
    "sar_box_mfma_flops": (_i64, [_i, _i, _i]),
    "sar_box_copy_f32": (_i, [_fp, _fp, _i64, _fp]),
    # host-side input helpers (host pointers)
    "sar_crc32c": (C.c_uint32, [_fp, _i64]),
    "sar_crc32c_sw": (C.c_uint32, [_fp, _i64]),
    "sar_masked_crc32c": (C.c_uint32, [_fp, _i64]),
    "sar_tfrecord_index": (_i64, [_fp, _i64, _i, _fp, _fp, _i64]),
}

_lib = None


class SarError(RuntimeError):
    pass


def load():
    """Load libsar_hip.so; raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SarError(
            "libsar_hip.so not found at %s -- build it with `python __graft_entry__.py build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    if (lib.sar_struct_size(0) != C.sizeof(ConvDesc) or lib.sar_struct_size(1) != C.sizeof(WgradDesc)
            or lib.sar_struct_size(2) != C.sizeof(Conv2dDesc)):
        raise SarError("descriptor layout mismatch between include/sar_hip.h and sar_amd/_lib.py")
    _lib = lib
    return lib


class Context:
    """A caller-owned sar_context (include/sar_hip.h): side streams + events on the CURRENT device for the launches that
    fan out (the parity classes of the 3x3 / stride-2 data gradient).  One per engine -- i.e. per host thread and device;
    the library itself holds no streams."""

    def __init__(self):
        h = C.c_void_p()
        check(load().sar_context_create(C.byref(h)), "sar_context_create")
        self.handle = h

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h and _lib is not None:
            try:
                _lib.sar_context_destroy(h)
            except Exception:
                pass


def check(rc, what=""):
    if rc != 0:
        msg = load().sar_last_error_string().decode("utf-8", "replace")
        raise SarError("%s failed (rc=%d): %s" % (what or "libsar_hip call", rc, msg))


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()


_raw_stream = None


def stream_ptr():
    """the current HIP stream of the current device as an integer handle.  torch.cuda.current_stream() builds a Stream object through
    four Python layers (~3 us; a Path B step asks 220 times: a third of the host's time per step, tools/host_profile.py): the raw
    query of torch's C module is used where it exists."""
    global _raw_stream
    import torch
    if _raw_stream is None:
        get_raw, get_dev = getattr(torch._C, "_cuda_getCurrentRawStream", None), getattr(torch._C, "_cuda_getDevice", None)
        if get_raw is not None and get_dev is not None:
            _raw_stream = lambda: get_raw(get_dev())
        else:
            _raw_stream = lambda: torch.cuda.current_stream().cuda_stream
    return _raw_stream()
