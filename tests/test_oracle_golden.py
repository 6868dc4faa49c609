"""CPU: the C oracle (oracle/msda_oracle.c) against the golden vectors generated from the reference's own
pure-PyTorch function (tests/golden/make_golden.py) -- this is what pins the oracle.

Tolerances: f64 cases 1e-12 abs (the two implementations order their sums differently: the reference
sums corner products through grid_sample then over L*P); f32 cases 2e-6 abs on O(1) data.
The reference's own test accepts rtol=1e-2/atol=1e-3 in f32 (.../ops/test.py:59); we are far inside it.
"""
import numpy as np
import pytest

from oracle import msda_oracle as orc

FWD_CASES = ["fwd_pow2_L3_f32", "fwd_nonpow2_edges_f32", "fwd_L4_P8_D32_f32", "fwd_oddD_f64",
             "grad_small_f64", "grad_head_geom_f32"]
GRAD_CASES = ["fwd_oddD_f64", "grad_small_f64", "grad_head_geom_f32"]


def _tol(dtype):
    return 1e-12 if dtype == np.float64 else 2e-6


def test_kat_reference_test_py(golden):
    """The reference's own recipe (.../ops/test.py:24-59, seed 3) and the values quoted in SURVEY.md 8c."""
    g = golden("kat_test_py")
    survey_d1 = [0.0018993784, 0.0046028276, 0.0046711755, 0.0043843999,
                 0.0037950971, 0.0025127644, 0.0018444262, 0.0036346796]
    survey_d2 = [0.0041157920, 0.0047920826, 0.0049443180, 0.0043424969,
                 0.0053378092, 0.0023724204, 0.0051279613, 0.0065011843]
    for tag, survey in (("d1", survey_d1), ("d2", survey_d2)):
        v, loc, w = g[tag + "_value"], g[tag + "_loc"], g[tag + "_attn"]
        out32 = orc.forward(v, g["shapes"], g["starts"], loc, w, im2col_step=2)
        out64 = orc.forward(v.astype(np.float64), g["shapes"], g["starts"], loc.astype(np.float64),
                            w.astype(np.float64), im2col_step=2)
        assert out32.shape == (1, 2, 4)
        np.testing.assert_allclose(out32, g[tag + "_out_f32"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(out64, g[tag + "_out_f64"], rtol=0, atol=1e-15)
        np.testing.assert_allclose(out32.ravel(), survey, rtol=0, atol=5e-10)


@pytest.mark.parametrize("name", FWD_CASES)
def test_forward_matches_reference_fixture(golden, name):
    g = golden(name)
    out = orc.forward(g["value"], g["shapes"], g["starts"], g["loc"], g["attn"])
    assert out.dtype == g["out"].dtype and out.shape == g["out"].shape
    np.testing.assert_allclose(out, g["out"], rtol=0, atol=_tol(out.dtype))


@pytest.mark.parametrize("name", GRAD_CASES)
def test_backward_matches_reference_autograd(golden, name):
    g = golden(name)
    gv, gl, ga = orc.backward(g["value"], g["shapes"], g["starts"], g["loc"], g["attn"], g["grad_out"])
    t = 1e-11 if gv.dtype == np.float64 else 2e-5
    np.testing.assert_allclose(gv, g["grad_value"], rtol=0, atol=t)
    np.testing.assert_allclose(ga, g["grad_attn"], rtol=0, atol=t)
    # d/dloc carries a factor W_l or H_l (<= 22 here)
    np.testing.assert_allclose(gl, g["grad_loc"], rtol=0, atol=t * 30)


def test_im2col_step_precondition():
    """batch % min(batch, im2col_step) == 0, else error (.../ops/src/cuda/ms_deform_attn_cuda.cu:55-57)."""
    rng = np.random.RandomState(0)
    shapes = np.array([[2, 2]], dtype=np.int64)
    starts = np.array([0], dtype=np.int64)
    v = rng.rand(3, 4, 1, 4).astype(np.float32)
    loc = rng.rand(3, 2, 1, 1, 1, 2).astype(np.float32)
    w = np.ones((3, 2, 1, 1, 1), dtype=np.float32)
    orc.forward(v, shapes, starts, loc, w, im2col_step=3)
    orc.forward(v, shapes, starts, loc, w, im2col_step=64)
    orc.forward(v, shapes, starts, loc, w, im2col_step=1)
    with pytest.raises(ValueError):
        orc.forward(v, shapes, starts, loc, w, im2col_step=2)


def test_properties_constant_value_and_integer_centres():
    """Domain properties the reference's semantics imply (cuh:38-89, 290-296):
    constant value + weights summing to 1 + all samples strictly inside -> constant output;
    a location at an exact pixel centre -> exact gather; a location outside the gate -> 0."""
    rng = np.random.RandomState(1)
    shapes = np.array([[5, 7], [3, 4]], dtype=np.int64)
    starts = np.array([0, 35], dtype=np.int64)
    S, M, D, Lq, L, P = 47, 2, 4, 6, 2, 3
    v = np.full((1, S, M, D), 2.5, dtype=np.float64)
    # strictly interior: pixel coords in [0, dim-1] <=> loc in [0.5/dim, 1-0.5/dim]
    loc = np.empty((1, Lq, M, L, P, 2))
    for l, (h, w) in enumerate(shapes):
        loc[:, :, :, l, :, 0] = rng.uniform(0.5 / w, 1 - 0.5 / w, size=(1, Lq, M, P))
        loc[:, :, :, l, :, 1] = rng.uniform(0.5 / h, 1 - 0.5 / h, size=(1, Lq, M, P))
    a = rng.rand(1, Lq, M, L, P)
    a /= a.sum((-1, -2), keepdims=True)
    out = orc.forward(v, shapes, starts, loc, a)
    np.testing.assert_allclose(out, 2.5, rtol=0, atol=1e-14)

    v = rng.randn(1, S, M, D)
    a = np.zeros((1, 1, M, L, P))
    a[..., 0, 0] = 1.0
    loc = np.zeros((1, 1, M, L, P, 2))
    loc[..., 0] = (3 + 0.5) / 7   # x = 3
    loc[..., 1] = (2 + 0.5) / 5   # y = 2
    out = orc.forward(v, shapes, starts, loc, a).reshape(M, D)
    np.testing.assert_allclose(out, v[0, 2 * 7 + 3], rtol=0, atol=1e-15)

    loc[...] = -0.2               # h_im = -0.2*5-0.5 = -1.5 -> gated out
    out = orc.forward(v, shapes, starts, loc, a)
    assert np.all(out == 0)
