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
