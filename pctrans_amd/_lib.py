"""ctypes binding of libpctrans_hip.so (C ABI: include/pctrans_hip.h).  Fails loudly if the library is missing."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libpctrans_hip.so")
ABI_VERSION = 1

_lib = None

_vp, _i = ctypes.c_void_p, ctypes.c_int
_FWD_ARGS = [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]
_BWD_ARGS = [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]

SYMBOLS = {
    "pct_abi_version": ([], _i),
    "pct_error_string": ([_i], ctypes.c_char_p),
    "pct_build_info": ([], ctypes.c_char_p),
    "pct_prepare_device": ([], _i),
    "pct_ms_deform_attn_forward_f32": (_FWD_ARGS, _i),
    "pct_ms_deform_attn_forward_f64": (_FWD_ARGS, _i),
    "pct_ms_deform_attn_forward_f16": (_FWD_ARGS, _i),
    "pct_ms_deform_attn_forward_bf16": (_FWD_ARGS, _i),
    "pct_ms_deform_attn_backward_f32": (_BWD_ARGS, _i),
    "pct_ms_deform_attn_backward_f64": (_BWD_ARGS, _i),
    "pct_ms_deform_attn_fused_forward_f32": ([_vp, _vp, _vp, _vp, ctypes.c_longlong, _vp, _vp] + [_i] * 7 + [_vp, _vp], _i),
    "pct_ms_deform_attn_forward_planes_f32": ([_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_longlong] + [_i] * 7 + [_vp, _vp], _i),
    "pct_add_layernorm_f32": ([_vp, _vp, _vp, _vp, ctypes.c_float, ctypes.c_longlong, _i, _vp, _vp], _i),
    "pct_linear_k128_f32": ([_vp, ctypes.c_longlong, _vp, ctypes.c_longlong, ctypes.c_longlong, _vp, _vp, ctypes.c_longlong, _i, _i, _vp,
                             ctypes.c_longlong, _vp], _i),
    "pct_msda_set_kernel_choice": ([_i], None),
    "pct_msda_last_kernel": ([], _i),
    "pct_msda_set_bwd_kernel_choice": ([_i], None),
    "pct_msda_last_bwd_kernel": ([], _i),
    "pct_linear_k128_multi_f32": ([_vp, ctypes.c_longlong, _vp, ctypes.c_longlong, ctypes.c_longlong, _i, _vp, _vp, _vp, _vp,
                                   _vp, _vp, ctypes.c_longlong, _vp], _i),
    "pct_linear_k128_add_layernorm_f32": ([_vp, ctypes.c_longlong, _vp, _vp, _vp, ctypes.c_longlong, _vp, _vp,
                                           ctypes.c_float, ctypes.c_longlong, _vp, ctypes.c_longlong, _vp], _i),
    "pct_linear_add_layernorm_f32": ([_vp, ctypes.c_longlong, _i, _vp, _vp, _vp, _vp, ctypes.c_longlong, _vp, _vp,
                                      ctypes.c_float, ctypes.c_longlong, _vp, ctypes.c_longlong, _vp], _i),
    "pct_groupnorm_flatten_f32": ([_vp, _vp, _vp, _i, _i, _i, _i, ctypes.c_float, _vp, _vp, ctypes.c_longlong,
                                   ctypes.c_longlong, _vp], _i),
    "pct_conv1x1_nchw_f32": ([_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp], _i),
    "pct_conv1x1_groupnorm_tokens_f32": ([_vp] * 6 + [_i, ctypes.c_float] + [_i] * 4 + [_vp, _vp, _vp, ctypes.c_longlong,
                                          ctypes.c_longlong, _vp], _i),
    "pct_ffn_layernorm_f32": ([_vp, ctypes.c_longlong, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_float, _i, ctypes.c_longlong, _vp, _vp,
                               ctypes.c_longlong, _vp], _i),
    "pct_lsap_f32": ([_vp, _i, _i, _i, _vp, _vp, _vp, _vp], _i),
    "pct_cross_attention_bf16": ([_vp] * 7 + [_i] * 4 + [ctypes.c_float, _vp, _vp], _i),
    "pct_masked_attention_bf16": ([_vp, _vp, _vp, _vp] + [_i] * 6 + [ctypes.c_float, _i, _vp, _vp], _i),
    "pct_dynamic_mask_head_forward_mfma": ([_vp, _vp, _vp] + [_i] * 9 + [_vp, _vp, _vp, _vp], _i),
    "pct_dynamic_mask_head_forward_fused_bf16": ([_vp, _vp, _vp] + [_i] * 9 + [_vp, _vp, _vp, _vp], _i),
    "pct_dynamic_mask_head_forward": ([_vp, _vp, _vp] + [_i] * 10 + [_vp, _vp, _vp], _i),
}


class PctransLibraryError(RuntimeError):
    pass


def lib():
    """Load (once) and return the HIP library.  No fallback: a missing library is an error."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PctransLibraryError(
            "libpctrans_hip.so not found at %s -- build it with `make -C pctrans_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`).  "
            "pctrans_amd has no CPU / eager fallback for device tensors." % LIB_PATH)
    l = ctypes.CDLL(LIB_PATH)
    for name, (argtypes, restype) in SYMBOLS.items():
        try:
            fn = getattr(l, name)
        except AttributeError as e:
            raise PctransLibraryError("libpctrans_hip.so does not export %s (stale build?)" % name) from e
        fn.argtypes = argtypes
        fn.restype = restype
    v = l.pct_abi_version()
    if v != ABI_VERSION:
        raise PctransLibraryError("libpctrans_hip.so ABI version %d, expected %d -- rebuild" % (v, ABI_VERSION))
    _lib = l
    return _lib


def check(code, what):
    if code != 0:
        msg = lib().pct_error_string(code).decode()
        raise RuntimeError("%s failed: %s (code %d)" % (what, msg, code))


_PREPARED = set()


def prepare_device(device):
    """Once per device: allocate the library's per-device pools now (include/pctrans_hip.h: pct_prepare_device), unless a
    stream capture is under way (the pools are then allocated by the first un-captured launch, as before)."""
    import torch
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx in _PREPARED:
        return
    if torch.cuda.is_current_stream_capturing():
        return
    with torch.cuda.device(idx):
        check(lib().pct_prepare_device(), "prepare_device")
    _PREPARED.add(idx)
