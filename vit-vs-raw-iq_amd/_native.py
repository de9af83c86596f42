"""ctypes binding of the C ABI declared in include/iqvit.h (libiqvit.so, gfx950).

There is no fallback: if the library is missing or a call fails the product path raises.
PyTorch is used above this layer only for device memory, streams and torch.distributed.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IQ_LIBIQVIT") or os.path.join(_HERE, "libiqvit.so")      # (override: A/B runs of two builds on one box)

STATUS = {0: "ok", 1: "invalid argument", 2: "unsupported shape/configuration", 3: "HIP launch error"}


class IqError(RuntimeError):
    pass


class Dropout(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("step", C.c_uint32), ("site", C.c_uint32), ("p", C.c_float),
                ("step_dev", C.c_void_p)]


class Epilogue(C.Structure):
    _fields_ = [("bias", C.c_void_p), ("relu", C.c_int), ("pe", C.c_void_p), ("tok", C.c_int), ("seq", C.c_int),
                ("cls_off", C.c_int), ("drop", Dropout), ("gate", C.c_void_p), ("ldg", C.c_int),
                ("gate_scale", C.c_float), ("residual", C.c_void_p), ("ldr", C.c_int)]


class WgradProblem(C.Structure):
    _fields_ = [("dY", C.c_void_p), ("ldy", C.c_int), ("X", C.c_void_p), ("ldx", C.c_int), ("dW", C.c_void_p),
                ("dbias", C.c_void_p), ("N", C.c_int), ("K", C.c_int)]


class ReduceSeg(C.Structure):
    _fields_ = [("partials", C.c_void_p), ("rows", C.c_int), ("row_stride", C.c_int64), ("out", C.c_void_p),
                ("n", C.c_int64)]


class ModelCfg(C.Structure):
    _fields_ = [("kind", C.c_int), ("in_channels", C.c_int), ("img_h", C.c_int), ("img_w", C.c_int),
                ("patch", C.c_int), ("seq_length", C.c_int), ("conv_k", C.c_int), ("use_cls", C.c_int),
                ("num_classes", C.c_int), ("d_model", C.c_int), ("n_head", C.c_int), ("n_layers", C.c_int),
                ("ffn_hidden", C.c_int), ("drop_prob", C.c_float)]


_P, _I, _Z, _F = C.c_void_p, C.c_int, C.c_size_t, C.c_float
_U64, _U32 = C.c_uint64, C.c_uint32

# name -> (restype, argtypes); must list every symbol include/iqvit.h declares
SIGNATURES = {
    "iq_ln_supported": (_I, [_I]),
    "iq_ln_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _F, _P]),
    "iq_ln_bwd_ws_bytes": (_Z, [_I]),
    "iq_ln_bwd_partial_rows": (_I, [_I, _I]),
    "iq_ln_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, C.POINTER(Dropout), _P, _P, _P, _I, _I, _I, _P]),
    "iq_gemm_bf16_nt": (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _I, C.POINTER(Epilogue), _P]),
    "iq_gemm_ln_supported": (_I, [_I, _I]),
    "iq_gemm_bf16_ln": (_I, [_P, _I, _P, _I, _P, _P, _I, C.POINTER(Dropout), _P, _P, _F, _P, _P, _P, _P, _I, _I, _I, _P]),
    "iq_gemm_lnbwd_supported": (_I, [_I, _I]),
    "iq_gemm_lnbwd_partial_rows": (_I, [_I]),
    "iq_gemm_bf16_lnbwd": (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _P, _P, C.POINTER(Dropout), _P, _P, _P, _I, _I, _I, _P]),
    "iq_ffn_chain_supported": (_I, [_I, _I, _I]),
    "iq_ffn_chain_bwd_partial_rows": (_I, [_I]),
    "iq_ffn_chain_bwd": (_I, [_P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _P, C.POINTER(Dropout), _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "iq_qkv_dgrad_ffn_chain_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, C.POINTER(Dropout), _P, _P, _P,
                                        _P, _P, _F, _P, _P, _P, _P, _P, _P, _P, C.POINTER(Dropout), _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "iq_ffn_chain_gate_bytes": (_Z, [_I, _I]),
    "iq_ffn_chain_fwd": (_I, [_P, _P, _P, C.POINTER(Dropout), _P, _P, _P, C.POINTER(Dropout), _P, _P, _F, _P, _P, _P, _P, _P, _I, _I,
                              _I, _I, _P]),
    "iq_attn_out_ffn_chain_fwd": (_I, [_P, _P, _P, C.POINTER(Dropout), _P, _P, _P, _P, _P, _P, _P,
                                       _P, _P, C.POINTER(Dropout), _P, _P, _P, C.POINTER(Dropout), _P, _P, _F, _P, _P, _P, _P, _P,
                                       _P, _P, _P, _I, _I, _I, _I, _P]),
    "iq_wgrad_ws_bytes": (_Z, [_I, _I, _I]),
    "iq_gemm_bf16_wgrad": (_I, [_P, _I, _P, _I, _P, _P, _I, _I, _I, _P, _Z, _I, _P]),
    "iq_wgrad_grouped_ws_bytes": (_Z, [_P, _I, _I, _I]),
    "iq_gemm_bf16_wgrad_grouped": (_I, [_P, _I, _I, _P, _Z, _I, _I, _P, _I, _P]),
    "iq_attn_supported": (_I, [_I, _I]),
    "iq_attn_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "iq_attn_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "iq_attn_fwd_masked": (_I, [_P, _P, _P, _P, C.c_long, _I, _I, _I, _I, _P]),
    "iq_attn_bwd_masked": (_I, [_P, _P, _P, _P, _P, _P, C.c_long, _I, _I, _I, _I, _P]),
    "iq_frames_preprocess": (_I, [_P, _P, _I, _I, _I, C.POINTER(C.c_float), _P]),
    "iq_patchify": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "iq_cls_rows": (_I, [_P, _P, _P, _I, _I, _I, C.POINTER(Dropout), _P]),
    "iq_embed_bwd_gather": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, C.POINTER(Dropout), _I, _P]),
    "iq_head_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "iq_ce_fwd_bwd": (_I, [_P, _P, _I, _I, _F, _F, _P, _P, _P, _P]),
    "iq_head_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "iq_gradnorm_ws_bytes": (_Z, [_Z]),
    "iq_gradnorm_sq": (_I, [_P, _Z, _F, _P, _P, _P]),
    "iq_adamw_step": (_I, [_P, _P, _P, _P, _P, _Z, _F, _F, _F, _F, _F, _I, _P, _F, _F, _P, _P]),
    "iq_counter_add": (_I, [_P, _U32, _P, _F, _P]),
    "iq_cast_bf16": (_I, [_P, _P, _Z, _P]),
    "iq_transpose_cast_bf16": (_I, [_P, _P, _I, _I, _P]),
    "iq_model_create": (_I, [C.POINTER(ModelCfg), C.POINTER(_P)]),
    "iq_model_destroy": (None, [_P]),
    "iq_model_last_error": (C.c_char_p, [_P]),
    "iq_model_tokens": (_I, [_P]),
    "iq_model_param_floats": (_Z, [_P]),
    "iq_model_param_entries": (_I, [_P]),
    "iq_model_param_entry": (_I, [_P, _I, C.c_char_p, _I, C.POINTER(_Z), C.POINTER(_I), C.POINTER(_I)]),
    "iq_model_shadow_bytes": (_Z, [_P]),
    "iq_model_workspace_bytes": (_Z, [_P, _I, _I]),
    "iq_model_bind": (_I, [_P, _P, _P, _P, _P]),
    "iq_model_bind_step_counter": (_I, [_P, _P]),
    "iq_model_refresh_shadow": (_I, [_P, _P]),
    "iq_model_refresh_transposed": (_I, [_P, _P]),
    "iq_model_forward": (_I, [_P, _P, _I, _P, _Z, _I, _U64, _U32, _P, _P, _P]),
    "iq_model_backward": (_I, [_P, _P, _P, _I, _P, _Z, _I, _I, _I, _P]),
    "iq_model_grad_range": (_I, [_P, _I, _I, C.POINTER(_Z), C.POINTER(_Z)]),
    "iq_prof_enable": (_I, [_I]),
    "iq_prof_collect": (_I, [C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
    "iq_prof_kernels": (_Z, [C.c_char_p, _Z, _I]),
}

_lib = None


def lib():
    """Load libiqvit.so (once).  Raises if it has not been built -- never falls back."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise IqError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C vit-vs-raw-iq_amd/csrc`). "
                "There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)      # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what, model=None):
    if rc != 0:
        detail = ""
        if model is not None:
            msg = lib().iq_model_last_error(model)
            if msg:
                detail = ": " + msg.decode()
        raise IqError(f"{what} failed: {STATUS.get(rc, rc)}{detail}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream_handle():
    import torch
    return torch.cuda.current_stream().cuda_stream
