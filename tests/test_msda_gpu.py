"""GPU parity tests proper: the HIP kernels, called through the C ABI (ctypes -> libpctrans_hip.so), against
  (1) the committed golden vectors generated from the reference's own PyTorch function,
  (2) the C oracle on seeded inputs at sizes it finishes in seconds,
  (3) size-independent properties at BASELINE.json's full sizes.

Tolerance (north_star: "within 1e-4 fp32"): fp32 forward |err| <= 1e-4 absolute on O(1) data (observed ~1e-6);
fp64 1e-11; fp16 / bf16 value paths are new capability: compared with the fp32 oracle run on the SAME 16-bit-rounded
value, error bounded by output rounding (2^-11 resp. 2^-8 relative to |out|, + accumulation slack).
"""
import numpy as np
import pytest
import torch

from msda_cases import make_case, starts_of
from oracle import msda_oracle as orc

pytestmark = pytest.mark.gpu

FWD_GOLDEN = ["fwd_pow2_L3_f32", "fwd_nonpow2_edges_f32", "fwd_L4_P8_D32_f32", "fwd_oddD_f64",
              "grad_small_f64", "grad_head_geom_f32"]
GRAD_GOLDEN = ["fwd_oddD_f64", "grad_small_f64", "grad_head_geom_f32"]


@pytest.fixture(scope="module")
def MSDA():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from pctrans_amd import MultiScaleDeformableAttention as m
    from pctrans_amd import _lib
    _lib.lib()   # fail loudly here if the HIP library is missing
    return m


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def run_fwd(MSDA, c, step=64):
    return MSDA.ms_deform_attn_forward(dev(c["value"]), dev(c["shapes"]), dev(c["starts"]), dev(c["loc"]),
                                       dev(c["attn"]), step).cpu().numpy()


def run_bwd(MSDA, c, grad_out, step=64):
    g = MSDA.ms_deform_attn_backward(dev(c["value"]), dev(c["shapes"]), dev(c["starts"]), dev(c["loc"]),
                                     dev(c["attn"]), dev(grad_out), step)
    return [t.cpu().numpy() for t in g]


def tol(dtype):
    return 1e-11 if dtype == np.float64 else 1e-4


# ---------------------------------------------------------------- golden vectors -------------------------------
def test_kat_reference_test_py(MSDA, golden):
    g = golden("kat_test_py")
    for tag in ("d1", "d2"):
        c = dict(value=g[tag + "_value"], shapes=g["shapes"], starts=g["starts"], loc=g[tag + "_loc"],
                 attn=g[tag + "_attn"])
        out = run_fwd(MSDA, c, step=2)
        assert out.shape == (1, 2, 4)
        np.testing.assert_allclose(out, g[tag + "_out_f32"], rtol=0, atol=1e-8)   # values are O(5e-3)
        c64 = {k: (v.astype(np.float64) if v.dtype == np.float32 else v) for k, v in c.items()}
        np.testing.assert_allclose(run_fwd(MSDA, c64, step=2), g[tag + "_out_f64"], rtol=0, atol=1e-15)


@pytest.mark.parametrize("name", FWD_GOLDEN)
def test_forward_golden(MSDA, golden, name):
    g = golden(name)
    out = run_fwd(MSDA, g)
    assert out.dtype == g["out"].dtype and out.shape == g["out"].shape
    np.testing.assert_allclose(out, g["out"], rtol=0, atol=tol(out.dtype))


@pytest.mark.parametrize("name", GRAD_GOLDEN)
def test_backward_golden(MSDA, golden, name):
    g = golden(name)
    gv, gl, ga = run_bwd(MSDA, g, g["grad_out"])
    t = 1e-10 if gv.dtype == np.float64 else 1e-4
    np.testing.assert_allclose(gv, g["grad_value"], rtol=0, atol=t)
    np.testing.assert_allclose(ga, g["grad_attn"], rtol=0, atol=t)
    np.testing.assert_allclose(gl, g["grad_loc"], rtol=0, atol=t * 30)     # carries a factor W_l / H_l <= 22


# ---------------------------------------------------------------- oracle on seeded inputs ----------------------
CASES = [
    # (id, kwargs)  -- BASELINE.json configs at sizes the oracle finishes in seconds
    ("cfg1_256_L3_Q=S", dict(seed=1, N=1, M=8, D=16, Lq=1344, P=4, shapes=[(8, 8), (16, 16), (32, 32)], model_like=True)),
    ("cfg2_512_L3_uniform", dict(seed=2, N=2, M=8, D=16, Lq=5376, P=4, shapes=[(16, 16), (32, 32), (64, 64)])),
    ("cfg2_512_L4_model", dict(seed=3, N=1, M=8, D=16, Lq=21760, P=4,
                               shapes=[(16, 16), (32, 32), (64, 64), (128, 128)], model_like=True)),
    ("cfg4_bbbc_nonpow2", dict(seed=4, N=2, M=8, D=16, Lq=7481, P=4, shapes=[(17, 22), (33, 44), (65, 87)],
                               model_like=True, px_sigma=3.0)),
    ("cfg3_cvppp_val", dict(seed=5, N=1, M=8, D=16, Lq=700, P=4, shapes=[(17, 16), (34, 32), (67, 63)],
                            lo=-0.1, hi=1.1)),
    ("cfg5_L5_P8", dict(seed=6, N=1, M=8, D=16, Lq=3000, P=8,
                        shapes=[(4, 4), (8, 8), (16, 16), (32, 32), (64, 64)], lo=-0.05, hi=1.05)),
    ("D32_M4", dict(seed=7, N=3, M=4, D=32, Lq=333, P=4, shapes=[(9, 7), (5, 3)])),
    ("D64_M2_P3", dict(seed=8, N=2, M=2, D=64, Lq=100, P=3, shapes=[(9, 7), (5, 3)])),
    ("D4_M1_single_level", dict(seed=9, N=1, M=1, D=4, Lq=77, P=1, shapes=[(3, 5)], lo=-0.5, hi=1.5)),
    ("D5_scalar_path", dict(seed=10, N=2, M=3, D=5, Lq=41, P=2, shapes=[(6, 4), (3, 2)], lo=-0.2, hi=1.2)),
    ("D1", dict(seed=11, N=1, M=2, D=1, Lq=19, P=2, shapes=[(2, 2), (1, 1)])),
    ("D12_cv3", dict(seed=12, N=1, M=5, D=12, Lq=64, P=4, shapes=[(7, 9), (4, 5)])),
    ("ragged_tail_block", dict(seed=13, N=1, M=3, D=16, Lq=23, P=4, shapes=[(5, 5)])),
]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("cid,kw", CASES, ids=[c[0] for c in CASES])
def test_forward_vs_oracle(MSDA, cid, kw, dtype):
    c = make_case(dtype=dtype, **kw)
    want = orc.forward(c["value"], c["shapes"], c["starts"], c["loc"], c["attn"])
    got = run_fwd(MSDA, c)
    np.testing.assert_allclose(got, want, rtol=0, atol=tol(dtype))


BWD_CASES = [c for c in CASES if c[0] in ("cfg1_256_L3_Q=S", "cfg4_bbbc_nonpow2", "cfg5_L5_P8", "D32_M4",
                                           "D64_M2_P3", "D4_M1_single_level", "D5_scalar_path", "D1", "D12_cv3",
                                           "ragged_tail_block", "cfg3_cvppp_val")]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("cid,kw", BWD_CASES, ids=[c[0] for c in BWD_CASES])
def test_backward_vs_oracle(MSDA, cid, kw, dtype):
    c = make_case(dtype=dtype, **kw)
    N, Lq = c["loc"].shape[:2]
    MD = c["value"].shape[2] * c["value"].shape[3]
    go = np.random.RandomState(kw["seed"] + 100).standard_normal((N, Lq, MD)).astype(dtype)
    want = orc.backward(c["value"], c["shapes"], c["starts"], c["loc"], c["attn"], go)
    got = run_bwd(MSDA, c, go)
    # grad_value is a sum of up to thousands of atomics in arbitrary order: relative-to-scale tolerance
    for g, w, name in zip(got, want, ("grad_value", "grad_loc", "grad_attn")):
        scale = max(1.0, float(np.abs(w).max()))
        t = (1e-10 if dtype == np.float64 else 2e-5) * scale
        np.testing.assert_allclose(g, w, rtol=0, atol=t, err_msg=name)


def test_autograd_function_matches_oracle(MSDA):
    from pctrans_amd.pixel_decoder.ops.functions import MSDeformAttnFunction
    c = make_case(seed=21, N=2, M=8, D=16, Lq=50, P=4, shapes=[(6, 7), (3, 4)], dtype=np.float64, lo=-0.1, hi=1.1)
    v, loc, a = dev(c["value"]).requires_grad_(), dev(c["loc"]).requires_grad_(), dev(c["attn"]).requires_grad_()
    out = MSDeformAttnFunction.apply(v, dev(c["shapes"]), dev(c["starts"]), loc, a, 2)
    go = torch.randn_like(out)
    out.backward(go)
    want = orc.backward(c["value"], c["shapes"], c["starts"], c["loc"], c["attn"], go.cpu().numpy())
    for g, w in zip((v.grad, loc.grad, a.grad), want):
        np.testing.assert_allclose(g.cpu().numpy(), w, rtol=0, atol=1e-10)


def test_gradcheck_like_reference_test_py(MSDA):
    """The reference's gradcheck recipe (OPS/test.py:66-89), channel counts hitting each of our lane layouts."""
    from torch.autograd import gradcheck
    from pctrans_amd.pixel_decoder.ops.functions import MSDeformAttnFunction
    N, M, Lq, L, P = 1, 2, 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long).cuda()
    starts = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    torch.manual_seed(3)
    for channels in (30, 32, 64, 71, 1025):
        value = (torch.rand(N, S, M, channels).cuda() * 0.01).double().requires_grad_()
        loc = torch.rand(N, Lq, M, L, P, 2).cuda().double().requires_grad_()
        w = torch.rand(N, Lq, M, L, P).cuda() + 1e-5
        w = (w / w.sum(-1, keepdim=True).sum(-2, keepdim=True)).double().requires_grad_()
        assert gradcheck(MSDeformAttnFunction.apply, (value, shapes, starts, loc, w, 2)), channels


# ---------------------------------------------------------------- 16-bit value paths (new capability) ----------
@pytest.mark.parametrize("tdt,eps", [(torch.float16, 2.0 ** -11), (torch.bfloat16, 2.0 ** -8)])
def test_forward_16bit_value(MSDA, tdt, eps):
    c = make_case(seed=31, N=2, M=8, D=16, Lq=5376, P=4, shapes=[(16, 16), (32, 32), (64, 64)], model_like=True)
    v16 = torch.from_numpy(c["value"]).to(tdt)
    c_round = dict(c, value=v16.float().numpy())
    want = orc.forward(c_round["value"], c["shapes"], c["starts"], c["loc"], c["attn"])
    got = MSDA.ms_deform_attn_forward(v16.cuda(), dev(c["shapes"]), dev(c["starts"]), dev(c["loc"]), dev(c["attn"]),
                                      64)
    assert got.dtype == tdt
    err = np.abs(got.float().cpu().numpy() - want)
    assert np.all(err <= eps * np.abs(want) + 1e-5), float(err.max())


# ---------------------------------------------------------------- edge cases & error behaviour -----------------
def test_empty_inputs(MSDA):
    shapes = torch.tensor([[2, 3]], dtype=torch.long).cuda()
    starts = torch.tensor([0], dtype=torch.long).cuda()
    v = torch.randn(2, 6, 2, 4).cuda()
    out = MSDA.ms_deform_attn_forward(v, shapes, starts, torch.zeros(2, 0, 2, 1, 3, 2).cuda(),
                                      torch.zeros(2, 0, 2, 1, 3).cuda(), 64)
    assert out.shape == (2, 0, 8)
    out = MSDA.ms_deform_attn_forward(v[:0].contiguous(), shapes, starts, torch.zeros(0, 5, 2, 1, 3, 2).cuda(),
                                      torch.zeros(0, 5, 2, 1, 3).cuda(), 64)
    assert out.shape == (0, 5, 8)


def test_error_behaviour(MSDA):
    c = make_case(seed=41, N=3, M=2, D=4, Lq=5, P=2, shapes=[(2, 2)])
    args = [dev(c[k]) for k in ("value", "shapes", "starts", "loc", "attn")]
    MSDA.ms_deform_attn_forward(*args, 3)
    MSDA.ms_deform_attn_forward(*args, 128)
    with pytest.raises(RuntimeError, match="im2col_step"):       # cu:57
        MSDA.ms_deform_attn_forward(*args, 2)
    with pytest.raises(RuntimeError, match="contiguous"):        # cu:33
        MSDA.ms_deform_attn_forward(args[0].transpose(2, 3).contiguous().transpose(2, 3), *args[1:], 3)
    with pytest.raises(RuntimeError, match="CPU"):               # ms_deform_attn.h:43
        MSDA.ms_deform_attn_forward(args[0].cpu(), *args[1:], 3)
    with pytest.raises(RuntimeError, match="CUDA tensor"):       # cu:39-43
        MSDA.ms_deform_attn_forward(args[0], args[1].cpu(), *args[2:], 3)


def test_nan_inf_in_unread_texels_do_not_leak(MSDA):
    """Corners outside the map read 0 in the reference (cuh:59-82): an Inf elsewhere in `value` must not appear."""
    shapes = np.array([[4, 4]], dtype=np.int64)
    v = np.zeros((1, 16, 1, 4), dtype=np.float32)
    v[0, 0] = np.inf            # texel (0,0): also the clamp target of masked-out corner loads
    v[0, 5] = 1.0               # texel (1,1)
    loc = np.array([(1.5 / 4, 1.5 / 4), (-0.05, 0.5), (3.9 / 4, 3.9 / 4), (np.nan, 0.5)], dtype=np.float32)
    loc = loc.reshape(1, 1, 1, 1, 4, 2)
    a = np.full((1, 1, 1, 1, 4), 0.25, dtype=np.float32)
    c = dict(value=v, shapes=shapes, starts=starts_of(shapes), loc=loc, attn=a)
    got = run_fwd(MSDA, c)
    want = orc.forward(v, shapes, c["starts"], loc, a)
    assert np.isfinite(got).all()
    np.testing.assert_allclose(got, want, atol=1e-6)


def test_unaligned_views_take_the_scalar_path(MSDA):
    """A contiguous view whose data_ptr is only 4-byte aligned must still be correct."""
    c = make_case(seed=51, N=1, M=2, D=8, Lq=33, P=4, shapes=[(5, 6), (3, 3)])
    want = orc.forward(c["value"], c["shapes"], c["starts"], c["loc"], c["attn"])
    flat = torch.zeros(c["value"].size + 1).cuda()
    flat[1:] = dev(c["value"]).flatten()
    v = flat[1:].view(c["value"].shape)
    assert v.is_contiguous() and v.data_ptr() % 16 != 0
    got = MSDA.ms_deform_attn_forward(v, dev(c["shapes"]), dev(c["starts"]), dev(c["loc"]), dev(c["attn"]), 64)
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=0, atol=1e-4)



# ---------------------------------------------------------------- windowed-LDS kernel (large problems) ---------
WIN_CASES = [
    # big enough (N*Lq*M >= 32768) to take the windowed kernel; pyramid mode (Lq == S) and flat mode (Lq != S)
    ("win_cfg2_L4_model", dict(seed=61, N=2, M=8, D=16, Lq=21760, P=4,
                               shapes=[(16, 16), (32, 32), (64, 64), (128, 128)], model_like=True)),
    ("win_cfg2_L3_uniform_all_fallback", dict(seed=62, N=2, M=8, D=16, Lq=5376, P=4,
                                              shapes=[(16, 16), (32, 32), (64, 64)])),
    ("win_bbbc_nonpow2_sigma6", dict(seed=63, N=2, M=8, D=16, Lq=7481, P=4, shapes=[(17, 22), (33, 44), (65, 87)],
                                     model_like=True, px_sigma=6.0)),
    ("win_edges_spill", dict(seed=64, N=1, M=8, D=16, Lq=5376, P=4, shapes=[(16, 16), (32, 32), (64, 64)],
                             lo=-0.3, hi=1.3)),
    ("win_flat_queries_Lq_ne_S", dict(seed=65, N=3, M=8, D=16, Lq=3000, P=4, shapes=[(20, 30), (10, 15)],
                                      lo=0.4, hi=0.6)),
    ("win_L5_P8", dict(seed=66, N=1, M=8, D=16, Lq=5456, P=8,
                       shapes=[(4, 4), (8, 8), (16, 16), (32, 32), (64, 64)], model_like=True, px_sigma=1.0)),
    ("win_M4_heads", dict(seed=67, N=4, M=4, D=16, Lq=2320, P=4, shapes=[(40, 50), (16, 20)], model_like=True)),
]


@pytest.fixture
def windowed(MSDA):
    """Force the windowed-LDS forward kernel: "auto" only takes it when the problem gives its persistent grid about
    three work items per workgroup, and these cases are kept small for the CPU oracle."""
    from pctrans_amd import _lib
    _lib.lib().pct_msda_set_kernel_choice(1)
    yield
    _lib.lib().pct_msda_set_kernel_choice(-1)


@pytest.mark.parametrize("cid,kw", WIN_CASES, ids=[c[0] for c in WIN_CASES])
def test_forward_windowed_kernel_vs_oracle(MSDA, windowed, cid, kw):
    c = make_case(dtype=np.float32, **kw)
    want = orc.forward(c["value"], c["shapes"], c["starts"], c["loc"], c["attn"])
    got = run_fwd(MSDA, c)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-4)


def test_windowed_kernel_tile_local_inf_does_not_leak(MSDA, windowed):
    """A non-finite texel inside a staged window must only reach outputs whose samples really read it."""
    c = make_case(seed=71, N=1, M=8, D=16, Lq=5376, P=4, shapes=[(16, 16), (32, 32), (64, 64)], model_like=True,
                  px_sigma=1.0)
    c["value"][0, 100, 3, :] = np.inf
    want = orc.forward(c["value"], c["shapes"], c["starts"], c["loc"], c["attn"])
    got = run_fwd(MSDA, c)
    np.testing.assert_array_equal(np.isfinite(got), np.isfinite(want))
    fin = np.isfinite(want)
    np.testing.assert_allclose(got[fin], want[fin], rtol=0, atol=1e-4)


@pytest.mark.parametrize("P,shapes", [(4, [(16, 16), (32, 32), (64, 64), (128, 128)]),
                                      (8, [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64)])])      # P = 8, L = 5: config 5
@pytest.mark.parametrize("tdt,eps", [(torch.float16, 2.0 ** -11), (torch.bfloat16, 2.0 ** -8)])
def test_windowed_kernel_16bit(MSDA, windowed, tdt, eps, P, shapes):
    Lq = sum(h * w for h, w in shapes)
    c = make_case(seed=72, N=2, M=8, D=16, Lq=Lq, P=P, shapes=shapes, model_like=True)
    v16 = torch.from_numpy(c["value"]).to(tdt)
    want = orc.forward(v16.float().numpy(), c["shapes"], c["starts"], c["loc"], c["attn"])
    got = MSDA.ms_deform_attn_forward(v16.cuda(), dev(c["shapes"]), dev(c["starts"]), dev(c["loc"]), dev(c["attn"]),
                                      64)
    err = np.abs(got.float().cpu().numpy() - want)
    assert np.all(err <= eps * np.abs(want) + 1e-5), float(err.max())


BWD_WIN_CASES = [c for c in WIN_CASES if c[1]["P"] == 4]


@pytest.mark.parametrize("cid,kw", BWD_WIN_CASES, ids=[c[0] for c in BWD_WIN_CASES])
def test_backward_windowed_kernel_vs_oracle(MSDA, cid, kw):
    """Large pyramid-mode problems take msda_backward_win.hip (LDS-window accumulation, one flush per window); the
    uniform / spilled cases exercise its per-level direct path, the Lq != S case the generic kernel."""
    c = make_case(dtype=np.float32, **kw)
    N, Lq = c["loc"].shape[:2]
    go = np.random.RandomState(kw["seed"] + 100).standard_normal((N, Lq, 8 * 16 if kw["M"] == 8 else kw["M"] * 16))
    go = go.astype(np.float32)
    want = orc.backward(c["value"], c["shapes"], c["starts"], c["loc"], c["attn"], go)
    got = run_bwd(MSDA, c, go)
    # grad_loc is discontinuous where a sample sits on a cell boundary (floor() flips): the kernel forms loc*W-0.5
    # with one FMA, the oracle with two roundings, so samples within 1e-3 px of a boundary are not compared
    wh = np.stack([c["shapes"][:, 1], c["shapes"][:, 0]], -1).astype(np.float64)          # (W, H) per level
    pix = c["loc"].astype(np.float64) * wh[None, None, None, :, None, :] - 0.5
    on_edge = (np.abs(pix - np.round(pix)) < 1e-3).any(-1, keepdims=True)
    assert on_edge.mean() < 0.01
    for g, w, name in zip(got, want, ("grad_value", "grad_loc", "grad_attn")):
        scale = max(1.0, float(np.abs(w).max()))
        if name == "grad_loc":
            g, w = np.where(on_edge, 0, g), np.where(on_edge, 0, w)
        np.testing.assert_allclose(g, w, rtol=0, atol=2e-5 * scale, err_msg=name)


@pytest.mark.parametrize("go_scale", [1e-20, 1.0, 1e20])
def test_backward_windowed_fixed_point_follows_the_gradient_scale(MSDA, go_scale):
    """The LDS accumulators are int32 fixed point with a per-tile power-of-two scale taken from max|grad_out| *
    max|attn|: the error stays relative to the gradient's own magnitude over 40 orders of magnitude."""
    c = make_case(seed=74, N=1, M=8, D=16, Lq=5376, P=4, shapes=[(16, 16), (32, 32), (64, 64)], model_like=True)
    go = (np.random.RandomState(174).standard_normal((1, 5376, 128)) * go_scale).astype(np.float32)
    want = orc.backward(c["value"], c["shapes"], c["starts"], c["loc"], c["attn"], go)
    got = run_bwd(MSDA, c, go)
    for g, w, name in zip(got, want, ("grad_value", "grad_attn")):
        np.testing.assert_allclose(g / go_scale, w / go_scale, rtol=0,
                                   atol=2e-5 * max(1.0, float(np.abs(w / go_scale).max())), err_msg=name)


def test_backward_windowed_nonfinite_grad_out_takes_the_float_path(MSDA):
    """A tile whose grad_out holds Inf / NaN cannot be scaled to fixed point: it must produce what float atomics do."""
    c = make_case(seed=75, N=1, M=8, D=16, Lq=5376, P=4, shapes=[(16, 16), (32, 32), (64, 64)], model_like=True)
    go = np.random.RandomState(175).standard_normal((1, 5376, 128)).astype(np.float32)
    go[0, 3000, 17] = np.inf
    go[0, 4100, 90] = np.nan
    want = orc.backward(c["value"], c["shapes"], c["starts"], c["loc"], c["attn"], go)
    got = run_bwd(MSDA, c, go)
    for g, w, name in zip(got, want, ("grad_value", "grad_loc", "grad_attn")):
        np.testing.assert_array_equal(np.isnan(g), np.isnan(w), err_msg=name)
        np.testing.assert_array_equal(np.isposinf(g), np.isposinf(w), err_msg=name)
        np.testing.assert_array_equal(np.isneginf(g), np.isneginf(w), err_msg=name)
        fin = np.isfinite(w)
        np.testing.assert_allclose(g[fin], w[fin], rtol=0, atol=2e-5 * max(1.0, float(np.abs(w[fin]).max())),
                                   err_msg=name)


def test_backward_windowed_gated_out_and_nonfinite_neighbours(MSDA):
    """Samples outside the map get exactly-zero gradients, and an Inf texel only reaches the gradients of samples
    that read it (the window's zero pixels / apron never inject 0 * Inf)."""
    c = make_case(seed=73, N=1, M=8, D=16, Lq=5376, P=4, shapes=[(16, 16), (32, 32), (64, 64)], model_like=True,
                  px_sigma=1.5)
    c["loc"][0, ::7, :, :, 1, :] = -3.0                      # far outside: gated out
    c["value"][0, 2000, 5, :] = np.inf
    go = np.random.RandomState(173).standard_normal((1, 5376, 128)).astype(np.float32)
    want = orc.backward(c["value"], c["shapes"], c["starts"], c["loc"], c["attn"], go)
    got = run_bwd(MSDA, c, go)
    assert np.all(got[1][0, ::7, :, :, 1, :] == 0) and np.all(got[2][0, ::7, :, :, 1] == 0)
    for g, w, name in zip(got, want, ("grad_value", "grad_loc", "grad_attn")):
        np.testing.assert_array_equal(np.isfinite(g), np.isfinite(w), err_msg=name)
        fin = np.isfinite(w)
        scale = max(1.0, float(np.abs(w[fin]).max()))
        np.testing.assert_allclose(g[fin], w[fin], rtol=0, atol=2e-5 * scale, err_msg=name)


# ---------------------------------------------------------------- full-size properties -------------------------
FULL = dict(N=8, M=8, D=16, P=4, shapes=[(16, 16), (32, 32), (64, 64), (128, 128)])   # north-star shape, Lq = S


def test_fullsize_constant_value_gives_constant_output(MSDA):
    shapes = np.asarray(FULL["shapes"], dtype=np.int64)
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    g = torch.Generator(device="cuda").manual_seed(0)
    N, M, D, P, L = FULL["N"], FULL["M"], FULL["D"], FULL["P"], len(shapes)
    v = torch.full((N, S, M, D), 1.75, device="cuda")
    # strictly interior samples: pixel coordinate in [0, dim-1]
    lo = torch.tensor([[0.5 / w, 0.5 / h] for h, w in shapes], device="cuda").view(1, 1, 1, L, 1, 2)
    loc = lo + torch.rand(N, S, M, L, P, 2, device="cuda", generator=g) * (1 - 2 * lo)
    a = torch.rand(N, S, M, L, P, device="cuda", generator=g) + 1e-3
    a = a / a.sum((-1, -2), keepdim=True)
    out = MSDA.ms_deform_attn_forward(v, dev(shapes), dev(starts_of(shapes)), loc, a, 64)
    assert out.shape == (N, S, M * D)
    assert float((out - 1.75).abs().max()) < 1e-5


def test_fullsize_linearity_and_head_independence(MSDA):
    shapes = np.asarray(FULL["shapes"], dtype=np.int64)
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    g = torch.Generator(device="cuda").manual_seed(1)
    N, M, D, P, L = FULL["N"], FULL["M"], FULL["D"], FULL["P"], len(shapes)
    sh, st = dev(shapes), dev(starts_of(shapes))
    v1 = torch.randn(N, S, M, D, device="cuda", generator=g)
    v2 = torch.randn(N, S, M, D, device="cuda", generator=g)
    loc = torch.rand(N, S, M, L, P, 2, device="cuda", generator=g) * 1.2 - 0.1
    a = torch.rand(N, S, M, L, P, device="cuda", generator=g)
    f = lambda v: MSDA.ms_deform_attn_forward(v, sh, st, loc, a, 8)
    o1, o2, o12 = f(v1), f(v2), f(2 * v1 - 3 * v2)
    assert float((o12 - (2 * o1 - 3 * o2)).abs().max()) < 2e-4
    # changing head 3 of value changes only head 3 of the output
    v3 = v1.clone()
    v3[:, :, 3] += 1.0
    d = (f(v3) - o1).view(N, S, M, D).abs().amax((0, 1, 3))
    assert float(d[3]) > 0 and float(d[[0, 1, 2, 4, 5, 6, 7]].max()) == 0.0


def test_fullsize_pixel_centre_locations_are_an_exact_gather(MSDA):
    shapes = np.asarray(FULL["shapes"], dtype=np.int64)
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    N, M, D = 2, FULL["M"], FULL["D"]
    L, P = len(shapes), 1
    g = torch.Generator(device="cuda").manual_seed(2)
    v = torch.randn(N, S, M, D, device="cuda", generator=g)
    # every query looks at one random pixel centre on level 3 with weight 1, weight 0 elsewhere
    h, w = int(shapes[3, 0]), int(shapes[3, 1])
    px = torch.randint(0, w, (N, S, M), device="cuda", generator=g)
    py = torch.randint(0, h, (N, S, M), device="cuda", generator=g)
    loc = torch.full((N, S, M, L, P, 2), 0.5, device="cuda")
    loc[:, :, :, 3, 0, 0] = (px + 0.5) / w
    loc[:, :, :, 3, 0, 1] = (py + 0.5) / h
    a = torch.zeros(N, S, M, L, P, device="cuda")
    a[:, :, :, 3, 0] = 1.0
    out = MSDA.ms_deform_attn_forward(v, dev(shapes), dev(starts_of(shapes)), loc, a, 64).view(N, S, M, D)
    start3 = int(starts_of(shapes)[3])
    idx = (start3 + py * w + px)                                      # [N, S, M]
    want = torch.gather(v, 1, idx[..., None].expand(N, S, M, D))
    assert float((out - want).abs().max()) < 1e-5


# ---------------------------------------------------------------- fused front-end (softmax + locations in-kernel) ---
@pytest.mark.parametrize("shapes,P,N,shared_ref", [
    ([(16, 16), (32, 32), (64, 64), (128, 128)], 4, 2, True),
    ([(17, 22), (33, 44), (65, 87)], 4, 2, False),
    ([(4, 4), (8, 8), (16, 16), (32, 32), (64, 64)], 8, 1, True),
    ([(5, 7)], 4, 3, False),
])
def test_fused_front_end_equals_module_math(MSDA, shapes, P, N, shared_ref):
    """fused(value, ref, offsets, logits) == unfused op on (ref + offsets/(W,H), softmax(logits)) -- both through the C ABI,
    and == the oracle on the same host-computed locations/weights."""
    rng = np.random.RandomState(5)
    sh = np.asarray(shapes, dtype=np.int64)
    L, M, D = len(shapes), 8, 16
    S = int((sh[:, 0] * sh[:, 1]).sum())
    Lq = S
    value = rng.standard_normal((N, S, M, D)).astype(np.float32)
    offsets = (rng.standard_normal((N, Lq, M, L, P, 2)) * 3).astype(np.float32)
    logits = (rng.standard_normal((N, Lq, M, L * P)) * 2).astype(np.float32)
    ref = rng.random_sample((1 if shared_ref else N, Lq, L, 2)).astype(np.float32)
    t = lambda a: torch.from_numpy(a).cuda()
    ref_t = t(ref).expand(N, -1, -1, -1) if shared_ref else t(ref)
    got = MSDA.ms_deform_attn_fused_forward(t(value), t(sh), t(starts_of(sh)), ref_t, t(offsets), t(logits))
    # the module's own math on the device (OPS/modules/ms_deform_attn.py:102-109)
    norm = torch.stack([t(sh)[:, 1], t(sh)[:, 0]], -1)
    loc = ref_t[:, :, None, :, None, :] + t(offsets) / norm[None, None, None, :, None, :]
    w = torch.softmax(t(logits), -1).view(N, Lq, M, L, P)
    unfused = MSDA.ms_deform_attn_forward(t(value), t(sh), t(starts_of(sh)), loc.contiguous(), w.contiguous(), 64)
    assert float((got - unfused).abs().max()) <= 1e-4
    want = orc.forward(value, sh, starts_of(sh), loc.cpu().numpy(), w.cpu().numpy())
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=0, atol=1e-4)


def test_module_fused_path_matches_reference_module_golden(MSDA, golden):
    """MSDeformAttn module on the GPU (fused front-end kernel, no grad) against the golden vector produced by the
    reference's own module (tests/golden/module_L3_d128.npz)."""
    from pctrans_amd.pixel_decoder.ops.modules import MSDeformAttn
    g = golden("module_L3_d128")
    m = MSDeformAttn(d_model=128, n_levels=3, n_heads=8, n_points=4)
    m.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd.")}, strict=True)
    m = m.cuda().eval()
    args = [torch.from_numpy(g[k]).cuda() for k in ("query", "ref", "src", "shapes", "starts")]
    with torch.no_grad():
        out = m(*args)                                   # fused kernel
    np.testing.assert_allclose(out.cpu().numpy(), g["out"], rtol=0, atol=1e-4)
    out2 = m(*args)                                      # grad enabled -> unfused op + autograd function
    np.testing.assert_allclose(out2.detach().cpu().numpy(), g["out"], rtol=0, atol=1e-4)
    out2.sum().backward()
    assert m.value_proj.weight.grad is not None and torch.isfinite(m.sampling_offsets.weight.grad).all()


def test_fullsize_backward_adjoint_identities(MSDA):
    """At the north-star size (oracle too slow): the op is linear in `value` and in `attn`, so its backward is the
    adjoint:  <go, forward(value)> == <grad_value, value> == <grad_attn, attn>."""
    c = make_case(seed=81, model_like=True, Lq=21760, **FULL)
    v, loc, attn = dev(c["value"]), dev(c["loc"]), dev(c["attn"])
    sh, st = dev(c["shapes"]), dev(c["starts"])
    go = torch.randn(FULL["N"], 21760, 128, device="cuda", generator=torch.Generator("cuda").manual_seed(5))
    out = MSDA.ms_deform_attn_forward(v, sh, st, loc, attn, 64)
    gv, gl, ga = MSDA.ms_deform_attn_backward(v, sh, st, loc, attn, go, 64)
    lhs = (go.double() * out.double()).sum().item()
    # rounding noise of a 22M-term dot product scales with the sum of magnitudes, not with the (random-sign) sum
    tol = 2e-8 * (go.double().abs() * out.double().abs()).sum().item()       # ~100x the observed atomic-order noise
    assert abs((gv.double() * v.double()).sum().item() - lhs) <= tol
    assert abs((ga.double() * attn.double()).sum().item() - lhs) <= tol
    assert torch.isfinite(gl).all()


def test_fullsize_forward_is_bitwise_repeatable(MSDA):
    """The forward has no atomics: 20 launches on the same inputs must agree bit for bit.  (Caught a missing
    s_waitcnt before the windowed kernel's loop-header barrier: 1 launch in ~8 gathered from stale windows.)"""
    c = make_case(seed=82, model_like=True, Lq=21760, **FULL)
    v, loc, attn = dev(c["value"]), dev(c["loc"]), dev(c["attn"])
    sh, st = dev(c["shapes"]), dev(c["starts"])
    first = MSDA.ms_deform_attn_forward(v, sh, st, loc, attn, 64)
    for _ in range(20):
        again = MSDA.ms_deform_attn_forward(v, sh, st, loc, attn, 64)
        assert torch.equal(first, again)


def test_fullsize_forward_replays_from_a_hip_graph(MSDA):
    """The windowed kernel hands out work items through per-XCD counters that it resets itself; captured in a HIP graph
    the launch must replay any number of times with the eager result (a per-launch memset node did not survive replay:
    stale counters, out-of-range items)."""
    c = make_case(seed=83, model_like=True, Lq=21760, **FULL)
    v, loc, attn = dev(c["value"]), dev(c["loc"]), dev(c["attn"])
    sh, st = dev(c["shapes"]), dev(c["starts"])
    eager = MSDA.ms_deform_attn_forward(v, sh, st, loc, attn, 64)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        MSDA.ms_deform_attn_forward(v, sh, st, loc, attn, 64)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = MSDA.ms_deform_attn_forward(v, sh, st, loc, attn, 64)
        out2 = MSDA.ms_deform_attn_forward(v, sh, st, loc, attn, 64)
    for _ in range(4):
        out.zero_()
        out2.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, eager) and torch.equal(out2, eager)
    # and eager launches interleaved with replays still agree
    assert torch.equal(MSDA.ms_deform_attn_forward(v, sh, st, loc, attn, 64), eager)


def test_static_item_stride_path_matches(MSDA):
    """PCT_WIN_QUEUE=0 (also the fallback when the counter ring cannot be allocated): same results from the static stride."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, torch; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from msda_cases import make_case\n"
        "from pctrans_amd import MultiScaleDeformableAttention as MSDA\n"
        "import oracle.msda_oracle as orc\n"
        "c = make_case(seed=5, model_like=True, Lq=5376, N=4, M=8, D=16, P=4, shapes=[(16, 16), (32, 32), (64, 64)])\n"
        "d = lambda a: torch.from_numpy(a).cuda()\n"
        "args = [d(c[k]) for k in ('value', 'shapes', 'starts', 'loc', 'attn')]\n"
        "out = MSDA.ms_deform_attn_forward(*args, 64)\n"
        "want = orc.forward(c['value'], c['shapes'], c['starts'], c['loc'], c['attn'])\n"
        "assert abs(out.cpu().numpy() - want).max() < 2e-5\n"
        "go = torch.randn_like(out)\n"
        "gv, gl, ga = MSDA.ms_deform_attn_backward(*args, go, 64)\n"
        "rv, rl, ra = orc.backward(c['value'], c['shapes'], c['starts'], c['loc'], c['attn'], go.cpu().numpy())\n"
        "assert abs(gv.cpu().numpy() - rv).max() < 1e-4 and abs(ga.cpu().numpy() - ra).max() < 1e-4\n"
        "print('ok')\n" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, PCT_WIN_QUEUE="0", PCT_MSDA_KERNEL="win")     # (auto would pick the quad-owner kernel at this size)
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "ok" in res.stdout, res.stderr[-2000:]
