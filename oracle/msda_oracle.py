"""ctypes/numpy front-end of oracle/msda_oracle.c.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows the call contract of the reference's extension entry points
(`ms_deform_attn_forward/backward`, .../pixel_decoder/ops/src/vision.cpp:18-21,
.../ops/src/cuda/ms_deform_attn_cuda.cu:25-158): same argument order, same
`batch % min(batch, im2col_step) == 0` precondition, output `[N, Lq, M*D]`.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmsda_oracle.so")
_lib = None


def build(force=False):
    """Compile the C restatement with gcc (seconds)."""
    src = os.path.join(_HERE, "msda_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.msda_oracle_max_threads.restype = ctypes.c_int
        for sfx in ("f32", "f64"):
            getattr(_lib, "msda_oracle_forward_" + sfx).restype = ctypes.c_int
            getattr(_lib, "msda_oracle_backward_" + sfx).restype = ctypes.c_int
    return _lib


def max_threads():
    return int(lib().msda_oracle_max_threads())


def set_threads(n):
    lib().msda_oracle_set_threads(ctypes.c_int(int(n)))


def _prep(value, shapes, starts, loc, attn):
    value = np.asarray(value)
    dt = value.dtype
    if dt not in (np.float32, np.float64):
        raise TypeError("oracle handles float32/float64 only, got %s" % dt)
    value = np.ascontiguousarray(value)
    loc = np.ascontiguousarray(loc, dtype=dt)
    attn = np.ascontiguousarray(attn, dtype=dt)
    shapes = np.ascontiguousarray(shapes, dtype=np.int64)
    starts = np.ascontiguousarray(starts, dtype=np.int64)
    N, S, M, D = value.shape
    _, Lq, M2, L, P, two = loc.shape
    assert M2 == M and two == 2 and attn.shape == (N, Lq, M, L, P) and shapes.shape == (L, 2)
    assert int((shapes[:, 0] * shapes[:, 1]).sum()) == S, "spatial_shapes do not add up to S"
    return value, shapes, starts, loc, attn, (N, S, M, D, L, Lq, P), ("f32" if dt == np.float32 else "f64")


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def forward(value, shapes, starts, loc, attn, im2col_step=64):
    value, shapes, starts, loc, attn, dims, sfx = _prep(value, shapes, starts, loc, attn)
    N, S, M, D, L, Lq, P = dims
    out = np.empty((N, Lq, M * D), dtype=value.dtype)
    rc = getattr(lib(), "msda_oracle_forward_" + sfx)(
        _p(value), _p(shapes), _p(starts), _p(loc), _p(attn),
        N, S, M, D, L, Lq, P, int(im2col_step), _p(out))
    if rc != 0:
        raise ValueError("batch(%d) must divide im2col_step(%d)" % (N, min(N, im2col_step)))
    return out


def backward(value, shapes, starts, loc, attn, grad_out, im2col_step=64):
    value, shapes, starts, loc, attn, dims, sfx = _prep(value, shapes, starts, loc, attn)
    N, S, M, D, L, Lq, P = dims
    grad_out = np.ascontiguousarray(grad_out, dtype=value.dtype).reshape(N, Lq, M * D)
    gv = np.empty_like(value)
    gl = np.empty_like(loc)
    ga = np.empty_like(attn)
    rc = getattr(lib(), "msda_oracle_backward_" + sfx)(
        _p(value), _p(shapes), _p(starts), _p(loc), _p(attn), _p(grad_out),
        N, S, M, D, L, Lq, P, int(im2col_step), _p(gv), _p(gl), _p(ga))
    if rc != 0:
        raise ValueError("batch(%d) must divide im2col_step(%d)" % (N, min(N, im2col_step)))
    return gv, gl, ga
