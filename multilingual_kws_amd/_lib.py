"""ctypes loader for lib/libmkws_hip.so (C-ABI: include/mkws.h).  Fails loudly; no fallback."""
import ctypes
import itertools
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MKWS_LIB selects another build of the same library (e.g. the phase-timing build of tools/README.md)
LIB_PATH = os.environ.get("MKWS_LIB") or os.path.join(_HERE, "lib", "libmkws_hip.so")

MKWS_OK = 0
MKWS_ERR_EXCHANGE = -7
ABI_VERSION = 5


class MkwsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libmkws_hip error {code}: {msg}")
        self.code = code


class FrontendCfg(ctypes.Structure):
    """mkws_frontend_cfg (include/mkws.h)."""
    _fields_ = [
        ("sample_rate", ctypes.c_int32), ("window_size_ms", ctypes.c_int32),
        ("window_step_ms", ctypes.c_int32), ("num_channels", ctypes.c_int32),
        ("upper_band_limit", ctypes.c_float), ("lower_band_limit", ctypes.c_float),
        ("smoothing_bits", ctypes.c_int32), ("even_smoothing", ctypes.c_float),
        ("odd_smoothing", ctypes.c_float), ("min_signal_remaining", ctypes.c_float),
        ("enable_pcan", ctypes.c_int32), ("pcan_strength", ctypes.c_float),
        ("pcan_offset", ctypes.c_float), ("gain_bits", ctypes.c_int32),
        ("enable_log", ctypes.c_int32), ("scale_shift", ctypes.c_int32),
    ]


# every symbol include/mkws.h declares: (name, restype, argtypes)
_P, _I, _F, _SZ = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
_CFG = ctypes.POINTER(FrontendCfg)
SYMBOLS = [
    ("mkws_abi_version", _I, []),
    ("mkws_last_error", ctypes.c_char_p, []),
    ("mkws_build_arch", ctypes.c_char_p, []),
    ("mkws_frontend_default_cfg", None, [_CFG]),
    ("mkws_frontend_host_table", _I, [_CFG, _I, _P, _SZ]),
    ("mkws_frontend_num_frames", _I, [_CFG, _I]),
    ("mkws_frontend_create", _I, [_CFG, _I, ctypes.POINTER(_P)]),
    ("mkws_frontend_destroy", None, [_P]),
    ("mkws_frontend_forward_f32", _I, [_P, _P, _I, _I, _P, _P, _P]),
    ("mkws_frontend_forward_i16", _I, [_P, _P, _I, _I, _P, _P, _P]),
    ("mkws_frontend_stream_f32", _I, [_P, _P, _I, _I, _I, _P, _P, _I, _P]),
    ("mkws_embed_weight_count", _SZ, []),
    ("mkws_embed_weight_manifest", _I, [ctypes.c_char_p, _SZ]),
    ("mkws_embed_create", _I, [_P, _SZ, _I, ctypes.POINTER(_P)]),
    ("mkws_embed_destroy", None, [_P]),
    ("mkws_embed_forward", _I, [_P, _P, _I, _P, _P]),
    ("mkws_embed_set_option", _I, [_P, ctypes.c_char_p, _I]),
    ("mkws_embed_get_option", _I, [_P, ctypes.c_char_p]),
    ("mkws_embed_profile", _I, [_P, _P, _I, _I, _P, ctypes.c_char_p, _SZ, _P]),
    ("mkws_embed_forward_tap", _I, [_P, _P, _I, ctypes.c_char_p, _P, _SZ, _P]),
    ("mkws_augment_batch", _I, [_P, _P, _P, ctypes.c_int64, _P, _I, _I, _P, _P]),
    ("mkws_specaug_apply", _I, [_P, _P, _I, _I, _I, _P]),
    ("mkws_specaug_apply_n", _I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    ("mkws_head_create", _I, [_I, _I, _I, _I, ctypes.POINTER(_P)]),
    ("mkws_head_destroy", None, [_P]),
    ("mkws_head_param_count", _I, [_P]),
    ("mkws_head_params", _P, [_P]),
    ("mkws_head_grads", _P, [_P]),
    ("mkws_head_state_floats", _I, [_P]),
    ("mkws_head_grad_count", _I, [_P]),
    ("mkws_head_set_params", _I, [_P, _P, _I, _P]),
    ("mkws_head_get_params", _I, [_P, _P, _I, _P]),
    ("mkws_head_forward", _I, [_P, _P, _I, _P, _P]),
    ("mkws_heads_forward", _I, [ctypes.POINTER(_P), _I, _P, _I, _P, _P]),
    ("mkws_head_loss_grad", _I, [_P, _P, _P, _I, _P, _P]),
    ("mkws_head_adam_step", _I, [_P, _F, _F, _F, _F, _I, _F, _P]),
    ("mkws_head_input_grad", _I, [_P, _P, _I, _P]),
    ("mkws_head_adam_step_dev", _I, [_P, _F, _F, _F, _F, _P, _F, _P]),
    # training operators (backprop_into_embedding)
    ("mkws_train_ctx_create", _I, [_P, _SZ, ctypes.POINTER(_P)]),
    ("mkws_train_ctx_destroy", None, [_P]),
    ("mkws_train_ctx_bind", _I, [_P]),
    ("mkws_op_set_scratch", _I, [_P, _SZ]),
    ("mkws_op_stream_wait", _I, [_P, _P]),
    ("mkws_op_set_option", _I, [ctypes.c_char_p, _I]),
    ("mkws_op_get_option", _I, [ctypes.c_char_p]),
    ("mkws_op_fold_defer", _I, [_I, _P]),
    ("mkws_op_fold_flush", _I, [_P]),
    ("mkws_op_bn_train_fwd", _I, [_P, _I, _I, _P, _P, _F, _I, _F, _P, _P, _P, _P, _P, _P]),
    ("mkws_op_bn_train_fwd_res", _I, [_P, _I, _I, _P, _P, _F, _I, _F, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    ("mkws_op_step_inc", _I, [_P, _P]),
    ("mkws_op_softmax_ce", _I, [_P, _P, _I, _I, _P, _P, _P]),
    ("mkws_op_dense_fwd", _I, [_P, _P, _P, _I, _P, _P, _I, _I, _I, _P]),
    ("mkws_op_adam_dev", _I, [_P, _P, _P, _P, ctypes.c_int64, _F, _F, _F, _F, _P, _F, _P]),
    ("mkws_op_gemm", _I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    ("mkws_op_bn_stats", _I, [_P, _I, _I, _P, _P, _P]),
    ("mkws_op_bn_act_fwd", _I, [_P, _P, _P, _P, _P, _F, _I, _P, _I, _I, _P]),
    ("mkws_op_bn_act_bwd", _I, [_P, _P, _P, _P, _P, _F, _I, _P, _P, _P, _P, _I, _I, _P]),
    ("mkws_op_bn_act_bwd_ex", _I, [_P, _P, _P, _P, _P, _F, _I, _P, _P, _P, _P, _F, _I, _P, _P, _I, _I, _P]),
    ("mkws_op_bn_update_moving", _I, [_P, _P, _P, _P, _F, _I, _I, _P]),
    ("mkws_op_dwconv_fwd", _I, [_P, _P, _P] + [_I] * 10 + [_P]),
    ("mkws_op_conv_bn_fwd", _I, [_P, _P, _P, _I, _I, _I, _P, _P, _F, _I, _F, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    ("mkws_op_dwconv_bn_fwd", _I, [_P, _P, _P] + [_I] * 10 + [_P, _P, _F, _I, _F, _P, _P, _P, _P, _P, _P]),
    ("mkws_op_dwconv_bwd", _I, [_P, _P, _P, _P, _P] + [_I] * 10 + [_P]),
    ("mkws_op_stem_fwd", _I, [_P, _P, _F, _F, _P, _I, _P]),
    ("mkws_op_stem_bwd_weight", _I, [_P, _P, _F, _F, _P, _I, _P]),
    ("mkws_op_pool_hw", _I, [_P, _P, _I, _I, _I, _P]),
    ("mkws_op_scale_channels", _I, [_P, _P, _P, _I, _I, _I, _P]),
    ("mkws_op_se_bwd", _I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    ("mkws_op_se_fwd", _I, [_P] * 11 + [_I, _I, _I, _I, _P]),
    ("mkws_op_se_bwd_fused", _I, [_P] * 17 + [_I, _I, _I, _I, _P]),
    ("mkws_op_se_wgrad", _I, [_P] * 8 + [_I, _I, _I, _P]),
    ("mkws_op_add_bcast", _I, [_P, _P, _F, _I, _I, _I, _P]),
    ("mkws_op_bias_act_fwd", _I, [_P, _P, _I, _P, _I, _I, _P]),
    ("mkws_op_bias_act_bwd", _I, [_P, _P, _I, _P, _P, _I, _I, _P]),
    ("mkws_op_row_scale_add", _I, [_P, _P, _P, _P, _I, ctypes.c_int64, _P]),
    ("mkws_op_axpy", _I, [_P, _P, _F, ctypes.c_int64, _P]),
    ("mkws_op_adam", _I, [_P, _P, _P, _P, ctypes.c_int64, _F, _F, _F, _F, _I, _F, _P]),
]

_lib = None


def lib():
    """Returns the loaded library; raises (never falls back) if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C multilingual_kws_amd/csrc`).  multilingual_kws_amd has no CPU fallback.")
        try:
            # Load order matters: PyTorch brings its own HIP runtime.  If this library (linked against
            # /opt/rocm's libamdhip64) is loaded first, the process ends up with a runtime that sees no device
            # on the GPU box; with torch first both share torch's.  torch is required anyway (device memory).
            import torch  # noqa: F401
        except ImportError:
            pass
        L = ctypes.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)       # AttributeError if the .so does not export it
            fn.restype = res
            fn.argtypes = args
        got = L.mkws_abi_version()
        if got != ABI_VERSION:
            raise ImportError(f"{LIB_PATH} has ABI version {got}, this package binds version {ABI_VERSION}: rebuild it (make -C multilingual_kws_amd/csrc)")
        _lib = L
    return _lib


_generation = itertools.count(1)


def next_generation():
    """Process-unique serial for every handle wrapper: caches keyed on raw handle addresses add it, so that a handle destroyed and
    re-created at the same address is not mistaken for the old one."""
    return next(_generation)


def forget_graphs(obj):
    """Called by the wrappers' close(): drops captured graphs that point into the handle being destroyed."""
    import sys
    m = sys.modules.get("multilingual_kws_amd.embedding.batch_streaming_analysis")
    if m is not None:
        m._BatchGraph.forget(obj)


def check(code):
    if code < 0:
        raise MkwsError(code, lib().mkws_last_error().decode("utf-8", "replace"))
    return code


def current_stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
