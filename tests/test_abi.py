"""CPU: the C-ABI library builds/loads and exports every symbol include/pctrans_hip.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "pctrans_hip.h")).read()
    return sorted(set(re.findall(r"PCT_API\s+[\w\s\*]+?\b(pct_\w+)\s*\(", hdr)))


def test_header_declares_expected_entry_points():
    syms = _declared_symbols()
    for s in ("pct_abi_version", "pct_error_string", "pct_ms_deform_attn_forward_f32",
              "pct_ms_deform_attn_forward_f64", "pct_ms_deform_attn_forward_f16", "pct_ms_deform_attn_forward_bf16",
              "pct_ms_deform_attn_backward_f32", "pct_ms_deform_attn_backward_f64"):
        assert s in syms


def test_library_exports_every_declared_symbol():
    from pctrans_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build first: make -C pctrans_amd/csrc"
    l = ctypes.CDLL(_lib.LIB_PATH)
    for s in _declared_symbols():
        assert hasattr(l, s), "libpctrans_hip.so does not export " + s
    assert set(_lib.SYMBOLS) == set(_declared_symbols()), "ctypes table and header disagree"
    assert _lib.lib().pct_abi_version() == _lib.ABI_VERSION
    assert b"im2col_step" in _lib.lib().pct_error_string(-2)


def test_library_reports_what_it_was_built_with_and_is_not_an_experiment_build():
    """The kernels carry compile-time A/B switches, some of them knock-outs that give wrong results (timing experiments,
    tools/variant.sh).  Those only compile with -DPCT_EXPERIMENT_BUILD; the library the tests run on must not be one."""
    from pctrans_amd import _lib
    info = _lib.lib().pct_build_info().decode()
    assert info.startswith("experiment=0; target=gfx950; "), info
    m = re.search(r"col: KO=(\d+)", info)
    assert m and set(m.group(1)) == {"0"}, info
    m = re.search(r"bcol: KO=(\d+)", info)
    assert m and m.group(1) == "0", info
    assert "STAMP=0" in info


def test_knock_out_switches_do_not_compile_without_the_experiment_flag(tmp_path):
    import subprocess
    src = os.path.join(ROOT, "pctrans_amd", "csrc", "msda_forward_col.hip")
    cmd = ["/opt/rocm/bin/hipcc", "-std=c++20", "--offload-arch=gfx950", "-fsyntax-only", "-DPCT_COL_KO_NOGATHER=1",
           "-I", os.path.join(ROOT, "include"), src]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "PCT_EXPERIMENT_BUILD" in r.stderr, r.stderr[-400:]


def test_missing_library_fails_loudly(monkeypatch):
    from pctrans_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libpctrans_hip.so")
    with pytest.raises(_lib.PctransLibraryError):
        _lib.lib()


def test_cpu_tensors_raise_like_the_reference_extension():
    import torch
    from pctrans_amd import MultiScaleDeformableAttention as MSDA
    v = torch.zeros(1, 4, 1, 4)
    shapes = torch.tensor([[2, 2]])
    starts = torch.tensor([0])
    loc = torch.zeros(1, 1, 1, 1, 1, 2)
    w = torch.zeros(1, 1, 1, 1, 1)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        MSDA.ms_deform_attn_forward(v, shapes, starts, loc, w, 64)
    with pytest.raises(RuntimeError, match="contiguous"):
        MSDA.ms_deform_attn_forward(v.expand(2, 4, 1, 4).transpose(0, 1), shapes, starts, loc, w, 64)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under pctrans_amd/ may reference it."""
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "pctrans_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")) or f == "Makefile":
                txt = open(os.path.join(d, f)).read()
                if re.search(r"\boracle\b", txt) and not f.endswith(".md"):
                    for line in txt.splitlines():
                        if re.search(r"(import|from|include|CDLL|-l).*\boracle", line):
                            bad.append((f, line.strip()))
    assert not bad, bad


def test_argument_validation_returns_error_codes_without_touching_the_device():
    """Bad arguments are rejected on the host (no launch, so this runs without a GPU): the C ABI reports errors through
    return codes, never by printing (the reference printf()s launch failures, cuh:953-957)."""
    from pctrans_amd import _lib
    L = _lib.lib()
    BAD, UNSUP, ALIGN = -1, -4, -3
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p).value                    # 16-byte aligned host pointer: never dereferenced
    assert p % 16 == 0
    # linear_k128: negative rows, n not a multiple of 32, misaligned x, row stride below K
    assert L.pct_linear_k128_f32(p, 128, None, 0, 0, p, p, -1, 128, 0, p, 128, None) == BAD
    assert L.pct_linear_k128_f32(p, 128, None, 0, 0, p, p, 8, 100, 0, p, 100, None) == UNSUP
    assert L.pct_linear_k128_f32(p + 4, 128, None, 0, 0, p, p, 8, 128, 0, p, 128, None) == ALIGN
    assert L.pct_linear_k128_f32(p, 64, None, 0, 0, p, p, 8, 128, 0, p, 128, None) == BAD
    assert L.pct_linear_k128_f32(p, 128, p, 128, 8, p, p, 64, 128, 0, p, 128, None) == BAD      # add_period < 32
    assert L.pct_linear_k128_f32(p, 128, None, 0, 0, p, p, 0, 128, 0, p, 128, None) == 0        # empty: OK, no launch
    # fused output_proj + LayerNorm: missing residual
    assert L.pct_linear_k128_add_layernorm_f32(p, 128, p, p, None, 128, p, p, 1e-5, 8, p, 128, None) == BAD
    # GroupNorm + flatten: channels other than 128, groups that do not divide into multiples of 4 channels
    assert L.pct_groupnorm_flatten_f32(p, p, p, 1, 64, 16, 32, 1e-5, p, p, 2048, 0, None) == UNSUP
    assert L.pct_groupnorm_flatten_f32(p, p, p, 1, 128, 16, 64, 1e-5, p, p, 2048, 0, None) == UNSUP
    assert L.pct_groupnorm_flatten_f32(p, p, p, 0, 128, 16, 32, 1e-5, p, p, 2048, 0, None) == 0
    # add + LayerNorm: unsupported width
    assert L.pct_add_layernorm_f32(p, None, p, p, 1e-5, 4, 96, p, None) == UNSUP
    assert b"" != L.pct_error_string(UNSUP)
