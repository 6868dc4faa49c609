"""GPU parity of the pyramid-column MSDeformAttn BACKWARD kernel (pctrans_amd/csrc/msda_backward_col.hip).

`auto` launches it for PCTrans' encoder geometry (fp32, Lq == S, D = 16, 4 points) once the call fills the persistent
grid; the oracle-sized cases below force it through the diagnostic switch of the C ABI and assert -- with
`pct_msda_last_bwd_kernel` -- that it really ran.  Reference semantics: ops/src/cuda/ms_deform_im2col_cuda.cuh:92-164,
:306-408, checked through the C oracle (oracle/msda_oracle.c).  Tolerance: 2e-5 of each gradient's own magnitude (the
LDS accumulators are fixed point with 2^-21 of the item's bound per contribution; observed <= 7e-6 on grad_value).
"""
import numpy as np
import pytest
import torch

from msda_cases import make_case
from oracle import msda_oracle as orc
from test_msda_col_gpu import COL_CASES, P1, P2, n_px

pytestmark = pytest.mark.gpu

B_WIN, B_GENERIC, B_COL = 1, 2, 3


@pytest.fixture(scope="module")
def MSDA():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from pctrans_amd import MultiScaleDeformableAttention as m
    from pctrans_amd import _lib
    _lib.lib()
    return m


@pytest.fixture(scope="module")
def lib():
    from pctrans_amd import _lib
    return _lib.lib()


def force(lib, k):
    class _Ctx:
        def __enter__(self):
            lib.pct_msda_set_bwd_kernel_choice(k)

        def __exit__(self, *a):
            lib.pct_msda_set_bwd_kernel_choice(-1)
    return _Ctx()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def run_bwd(MSDA, lib, c, go, kernel=B_COL):
    with force(lib, kernel):
        g = MSDA.ms_deform_attn_backward(dev(c["value"]), dev(c["shapes"]), dev(c["starts"]), dev(c["loc"]),
                                         dev(c["attn"]), dev(go), 64)
        torch.cuda.synchronize()
        assert lib.pct_msda_last_bwd_kernel() == kernel
    return [t.cpu().numpy() for t in g]


def grad_out_for(c, seed, scale=1.0):
    N, Lq, M = c["loc"].shape[:3]
    D = c["value"].shape[-1]
    return (np.random.RandomState(seed).standard_normal((N, Lq, M * D)) * scale).astype(np.float32)


def edge_mask(c, eps=1e-3):
    """grad_loc is discontinuous where a sample sits on a cell boundary (floor() flips): the kernel forms loc * W - 0.5
    with one FMA, the oracle with two roundings, so samples within eps px of a boundary are not compared."""
    wh = np.stack([c["shapes"][:, 1], c["shapes"][:, 0]], -1).astype(np.float64)          # (W, H) per level
    pix = c["loc"].astype(np.float64) * wh[None, None, None, :, None, :] - 0.5
    return (np.abs(pix - np.round(pix)) < eps).any(-1, keepdims=True)


def compare(got, want, c, atol_rel=2e-5, edge_eps=1e-3):
    on_edge = edge_mask(c, edge_eps)
    for g, w, name in zip(got, want, ("grad_value", "grad_loc", "grad_attn")):
        scale = max(1.0, float(np.abs(w).max()))
        if name == "grad_loc":
            g, w = np.where(on_edge, 0, g), np.where(on_edge, 0, w)
        np.testing.assert_allclose(g, w, rtol=0, atol=atol_rel * scale, err_msg=name)


BWD_COL_CASES = [c for c in COL_CASES if "long_strip" not in c[0] and "wider_than" not in c[0]]


@pytest.mark.parametrize("cid,kw", BWD_COL_CASES, ids=[c[0] for c in BWD_COL_CASES])
def test_backward_column_kernel_vs_oracle(MSDA, lib, cid, kw):
    """One phase (init-like), two phases (model-like), levels on the direct path (wide boxes, uniform locations), ragged
    grids, 5 levels, 1 / 4 / 8 heads."""
    kw = dict(dict(M=8, D=16, P=4), **kw)
    kw.pop("atol", None)
    c = make_case(dtype=np.float32, **kw)
    go = grad_out_for(c, kw["seed"] + 100)
    want = orc.backward(c["value"], c["shapes"], c["starts"], c["loc"], c["attn"], go)
    got = run_bwd(MSDA, lib, c, go)
    if not kw.get("init_like"):                      # (init-like offsets are whole pixels: every sample sits on a boundary)
        assert edge_mask(c).mean() < 0.01
    compare(got, want, c)


@pytest.mark.parametrize("cid,shapes", [("long_strip", [(1, 15000), (1, 30000), (1, 60000)]),
                                        ("wider_than_the_box_corners", [(1, 17500), (1, 35000), (1, 70000)])])
def test_backward_column_kernel_flat_columns_and_oversized_maps(MSDA, lib, cid, shapes):
    """Maps so elongated that the cell tables do not fit (flat columns of 256 consecutive queries) and maps wider than
    the 16-bit box corners (every level on the direct path).  On a map 60 000 pixels wide one ulp of a pixel coordinate is
    2^-8 px: samples within 2^-6 px of a cell boundary are left out of the grad_loc comparison, and the bilinear weights
    themselves carry that coordinate error (tolerance 2e-3 of the magnitude, as for the forward)."""
    c = make_case(seed=215, N=1, M=2, D=16, Lq=n_px(shapes), P=4, shapes=shapes, model_like=True, px_sigma=1.5)
    go = grad_out_for(c, 315)
    want = orc.backward(c["value"], c["shapes"], c["starts"], c["loc"], c["attn"], go)
    got = run_bwd(MSDA, lib, c, go)
    compare(got, want, c, atol_rel=2e-3, edge_eps=2.0 ** -6)


@pytest.mark.parametrize("go_scale", [1e-20, 1.0, 1e20])
def test_backward_column_fixed_point_follows_the_gradient_scale(MSDA, lib, go_scale):
    """The LDS accumulators are integers with one power-of-two scale per item from max|grad_out| * max|attn|: the error
    stays relative to the gradient's own magnitude over 40 orders of magnitude."""
    c = make_case(seed=74, N=1, M=8, D=16, Lq=n_px(P1), P=4, shapes=P1, model_like=True)
    go = grad_out_for(c, 174, go_scale)
    want = orc.backward(c["value"], c["shapes"], c["starts"], c["loc"], c["attn"], go)
    got = run_bwd(MSDA, lib, c, go)
    for g, w, name in zip(got, want, ("grad_value", "grad_loc", "grad_attn")):
        if name == "grad_loc":
            continue
        np.testing.assert_allclose(g / go_scale, w / go_scale, rtol=0,
                                   atol=2e-5 * max(1.0, float(np.abs(w / go_scale).max())), err_msg=name)


def test_backward_column_nonfinite_grad_out_takes_the_float_path(MSDA, lib):
    """An item whose grad_out holds Inf / NaN cannot be scaled to fixed point: it must produce what float atomics in the
    reference's summation order do (the same Inf / NaN pattern as the oracle)."""
    c = make_case(seed=75, N=1, M=8, D=16, Lq=n_px(P1), P=4, shapes=P1, model_like=True)
    go = grad_out_for(c, 175)
    go[0, 3000, 17] = np.inf
    go[0, 4100, 90] = np.nan
    want = orc.backward(c["value"], c["shapes"], c["starts"], c["loc"], c["attn"], go)
    got = run_bwd(MSDA, lib, c, go)
    for g, w, name in zip(got, want, ("grad_value", "grad_loc", "grad_attn")):
        np.testing.assert_array_equal(np.isnan(g), np.isnan(w), err_msg=name)
        np.testing.assert_array_equal(np.isposinf(g), np.isposinf(w), err_msg=name)
        np.testing.assert_array_equal(np.isneginf(g), np.isneginf(w), err_msg=name)
        fin = np.isfinite(w)
        np.testing.assert_allclose(g[fin], w[fin], rtol=0, atol=2e-5 * max(1.0, float(np.abs(w[fin]).max())),
                                   err_msg=name)


def test_backward_column_gated_out_and_nonfinite_neighbours(MSDA, lib):
    """Samples outside the map get exactly-zero gradients, and an Inf texel only reaches the gradients of samples that
    read it (the window's zero apron never injects 0 * Inf)."""
    c = make_case(seed=73, N=1, M=8, D=16, Lq=n_px(P1), P=4, shapes=P1, model_like=True, px_sigma=1.5)
    c["loc"][0, ::7, :, :, 1, :] = -3.0                      # far outside: gated out
    c["value"][0, 2000, 5, :] = np.inf
    go = grad_out_for(c, 173)
    want = orc.backward(c["value"], c["shapes"], c["starts"], c["loc"], c["attn"], go)
    got = run_bwd(MSDA, lib, c, go)
    assert np.all(got[1][0, ::7, :, :, 1, :] == 0) and np.all(got[2][0, ::7, :, :, 1] == 0)
    for g, w, name in zip(got, want, ("grad_value", "grad_loc", "grad_attn")):
        np.testing.assert_array_equal(np.isfinite(g), np.isfinite(w), err_msg=name)
        fin = np.isfinite(w)
        scale = max(1.0, float(np.abs(w[fin]).max()))
        np.testing.assert_allclose(g[fin], w[fin], rtol=0, atol=2e-5 * scale, err_msg=name)


def test_backward_column_is_bitwise_repeatable_and_matches_the_windowed_kernel(MSDA, lib):
    """Integer LDS sums are order-independent: two launches on the same inputs agree bit for bit in grad_loc / grad_attn,
    and grad_value differs only by the float atomics of the flush; the round-1 windowed kernel gives the same gradients."""
    c = make_case(seed=76, N=2, M=8, D=16, Lq=n_px(P2), P=4, shapes=P2, model_like=True)
    go = grad_out_for(c, 176)
    a = run_bwd(MSDA, lib, c, go)
    b = run_bwd(MSDA, lib, c, go)
    np.testing.assert_array_equal(a[1], b[1])
    np.testing.assert_array_equal(a[2], b[2])
    w = run_bwd(MSDA, lib, c, go, kernel=B_WIN)
    compare(a, w, c)


def test_backward_auto_takes_the_column_kernel_at_model_batch(MSDA, lib):
    """`auto` at the north-star shape (batch 8): the column kernel, and the adjoint identities of the op -- it is linear in
    `value` and in `attn`, so <grad_out, forward> = <grad_value, value> = <grad_attn, attn> (oracle too slow at this size)."""
    c = make_case(seed=77, N=8, M=8, D=16, Lq=n_px(P2), P=4, shapes=P2, model_like=True)
    go = grad_out_for(c, 177)
    v, sh, st, loc, attn = dev(c["value"]), dev(c["shapes"]), dev(c["starts"]), dev(c["loc"]), dev(c["attn"])
    gv, gl, ga = MSDA.ms_deform_attn_backward(v, sh, st, loc, attn, dev(go), 64)
    torch.cuda.synchronize()
    assert lib.pct_msda_last_bwd_kernel() == B_COL
    out = MSDA.ms_deform_attn_forward(v, sh, st, loc, attn, 64)
    lhs = float((out.double() * dev(go).double()).sum())
    r1 = float((gv.double() * v.double()).sum())
    r2 = float((ga.double() * attn.double()).sum())
    ref = float((out.double().abs() * dev(go).double().abs()).sum())
    assert abs(lhs - r1) <= 1e-5 * ref and abs(lhs - r2) <= 1e-5 * ref, (lhs, r1, r2, ref)


def test_backward_column_flag_overflow_route_matches():
    """More work items than the flag buffer holds: the main launch declines inside the kernel and the DIRECT launch does
    every level of every item (PCT_BCOL_FLAG_CAP shrinks the capacity so that an oracle-sized case gets there)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from msda_cases import make_case\n"
        "from oracle import msda_oracle as orc\n"
        "from pctrans_amd import MultiScaleDeformableAttention as MSDA, _lib\n"
        "P1 = [(16, 16), (32, 32), (64, 64)]\n"
        "c = make_case(seed=78, N=1, M=8, D=16, Lq=5376, P=4, shapes=P1, model_like=True)\n"
        "go = np.random.RandomState(178).standard_normal((1, 5376, 128)).astype(np.float32)\n"
        "dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()\n"
        "_lib.lib().pct_msda_set_bwd_kernel_choice(3)\n"
        "g = MSDA.ms_deform_attn_backward(dev(c['value']), dev(c['shapes']), dev(c['starts']), dev(c['loc']), dev(c['attn']), dev(go), 64)\n"
        "torch.cuda.synchronize()\n"
        "assert _lib.lib().pct_msda_last_bwd_kernel() == 3\n"
        "w = orc.backward(c['value'], c['shapes'], c['starts'], c['loc'], c['attn'], go)\n"
        "for a, b in ((g[0], w[0]), (g[2], w[2])):\n"
        "    np.testing.assert_allclose(a.cpu().numpy(), b, rtol=0, atol=2e-5 * max(1.0, float(np.abs(b).max())))\n"
        "print('ok')\n" % (root, os.path.join(root, "tests")))
    env = dict(os.environ, PCT_BCOL_FLAG_CAP="16")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("kernel,sigma", [(B_COL, 6.0), (B_COL, 2.0), (B_WIN, 2.0)], ids=["col_with_direct_levels", "col", "windowed"])
def test_backward_replays_from_a_hip_graph(MSDA, lib, kernel, sigma):
    """Captured in a HIP graph, the backward replays any number of times: the work queue's counters, the column kernel's flag
    bytes and `any` / `readers` words all reset themselves, a captured launch owns its flag buffer -- and grad_value is zeroed
    by a kernel of the library: a hipMemsetAsync recorded into the graph wrote garbage into every fourth element from the
    second replay on (ROCm 7.2), for the windowed kernel as for the column kernel.  sigma = 6 px: the finest level's boxes
    exceed the pool, so the column kernel's DIRECT launch has work."""
    c = make_case(seed=79, N=2, M=8, D=16, Lq=n_px(P2), P=4, shapes=P2, model_like=True, px_sigma=sigma)
    go = grad_out_for(c, 179)
    args = [dev(c["value"]), dev(c["shapes"]), dev(c["starts"]), dev(c["loc"]), dev(c["attn"]), dev(go), 64]
    with force(lib, kernel):
        eager = MSDA.ms_deform_attn_backward(*args)
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            MSDA.ms_deform_attn_backward(*args)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = MSDA.ms_deform_attn_backward(*args)
            assert lib.pct_msda_last_bwd_kernel() == kernel
        scale = float(eager[0].abs().max())
        for _ in range(4):
            for t in out:
                t.fill_(float("nan"))
            g.replay()
            torch.cuda.synchronize()
            if kernel == B_COL:                                                                # integer LDS sums: bitwise
                np.testing.assert_array_equal(out[1].cpu().numpy(), eager[1].cpu().numpy())
                np.testing.assert_array_equal(out[2].cpu().numpy(), eager[2].cpu().numpy())
            else:
                assert float((out[1] - eager[1]).abs().max()) <= 1e-5 * max(1.0, float(eager[1].abs().max()))
                assert float((out[2] - eager[2]).abs().max()) <= 1e-5 * max(1.0, float(eager[2].abs().max()))
            assert float((out[0] - eager[0]).abs().max()) <= 1e-5 * scale                      # grad_value: float atomics
        again = MSDA.ms_deform_attn_backward(*args)                                            # eager launches in between
        assert float((again[0] - eager[0]).abs().max()) <= 1e-5 * scale


def test_backward_launches_in_flight_on_several_streams_never_share_a_flag_buffer(MSDA, lib):
    """the eager flag-buffer ring has 8 slots.  Launch k and launch k + 8 used to share one whatever their streams;
    now a slot whose previous launch -- on ANOTHER stream -- has not completed is not handed out (that launch runs on the
    windowed kernel instead).  24 backward launches round-robin over 4 streams without any host synchronisation: every
    result equals the single-stream one (grad_sampling_loc / grad_attn_weight bit for bit for the launches that ran the
    column kernel; all within the oracle tolerance)."""
    c = make_case(seed=79, N=2, M=8, D=16, Lq=n_px(P2), P=4, shapes=P2, model_like=True, px_sigma=2.0)
    go = grad_out_for(c, 179)
    args = [dev(c["value"]), dev(c["shapes"]), dev(c["starts"]), dev(c["loc"]), dev(c["attn"]), dev(go), 64]
    with force(lib, B_COL):
        ref = MSDA.ms_deform_attn_backward(*args)
        torch.cuda.synchronize()
        assert lib.pct_msda_last_bwd_kernel() == B_COL
    streams = [torch.cuda.Stream() for _ in range(4)]
    for s in streams:
        s.wait_stream(torch.cuda.current_stream())
    results, kernels = [], []
    for i in range(24):
        with torch.cuda.stream(streams[i % 4]):
            results.append(MSDA.ms_deform_attn_backward(*args))          # auto: column kernel, or windowed when no slot is free
            kernels.append(lib.pct_msda_last_bwd_kernel())
    torch.cuda.synchronize()
    assert set(kernels) <= {B_COL, B_WIN} and kernels.count(B_COL) >= 8, kernels
    scale = [float(t.abs().max()) for t in ref]
    for k, out in zip(kernels, results):
        for j in range(3):
            assert float((out[j] - ref[j]).abs().max()) <= 2e-5 * scale[j], (k, j)
        if k == B_COL:
            assert torch.equal(out[1], ref[1]) and torch.equal(out[2], ref[2])
