"""GPU parity of the pyramid-column MSDeformAttn forward kernel (pctrans_amd/csrc/msda_forward_col.hip) and of the kernel
choice `auto` makes per BASELINE.json configuration.

The column kernel is what `auto` launches for PCTrans' encoder geometry (fp32, Lq == S, 4 points) once a call holds at
least two work items per CU, i.e. from a batch the CPU oracle no longer finishes quickly; the oracle-sized cases below
therefore force it through the diagnostic switch of the C ABI and assert -- with `pct_msda_last_kernel` -- that it
really ran.  Reference semantics: ops/src/cuda/ms_deform_im2col_cuda.cuh:242-304 (+ :38-89), checked through the C
oracle (oracle/msda_oracle.c); tolerance 1e-4 absolute on O(1) data (north_star), observed ~1e-6.
"""
import numpy as np
import pytest
import torch

from msda_cases import make_case, starts_of
from oracle import msda_oracle as orc

pytestmark = pytest.mark.gpu

K_WIN, K_GENERIC, K_DPP, K_COL = 1, 2, 3, 4
P2 = [(16, 16), (32, 32), (64, 64), (128, 128)]
P1 = [(16, 16), (32, 32), (64, 64)]
P4 = [(17, 22), (33, 44), (65, 87)]
P3 = [(16, 16), (32, 32), (64, 64), (128, 128), (256, 256)]


def n_px(shapes):
    return sum(h * w for h, w in shapes)


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
            lib.pct_msda_set_kernel_choice(k)

        def __exit__(self, *a):
            lib.pct_msda_set_kernel_choice(-1)
    return _Ctx()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def run_fwd(MSDA, c):
    return MSDA.ms_deform_attn_forward(dev(c["value"]), dev(c["shapes"]), dev(c["starts"]), dev(c["loc"]),
                                       dev(c["attn"]), 64).cpu().numpy()


COL_CASES = [
    # one phase: every level's box fits the pool together (what a random-init network gives: distribution I)
    ("col_P2_init_like", dict(seed=101, N=1, Lq=n_px(P2), shapes=P2, init_like=True)),
    ("col_P2_init_like_jitter", dict(seed=102, N=2, Lq=n_px(P2), shapes=P2, init_like=True, model_like=True, px_sigma=0.5)),
    # gaussian offsets, sigma = 2 px (distribution M): the finest level's box no longer fits beside the others -> 2 phases
    ("col_P2_model_sigma2", dict(seed=103, N=2, Lq=n_px(P2), shapes=P2, model_like=True)),
    ("col_P1_model_sigma2", dict(seed=104, N=2, Lq=n_px(P1), shapes=P1, model_like=True)),
    # wide boxes: several phases and levels gathered from global memory
    ("col_P4_nonpow2_sigma6", dict(seed=105, N=2, Lq=n_px(P4), shapes=P4, model_like=True, px_sigma=6.0)),
    ("col_P2_sigma12", dict(seed=106, N=1, Lq=n_px(P2), shapes=P2, model_like=True, px_sigma=12.0)),
    # uniform locations (ops/test.py:37): every box is the whole level; in / out of the map
    ("col_P1_uniform", dict(seed=107, N=2, Lq=n_px(P1), shapes=P1)),
    ("col_P1_edges", dict(seed=108, N=1, Lq=n_px(P1), shapes=P1, lo=-0.3, hi=1.3)),
    # ragged column grids: levels that do not divide, a level narrower than the grid, 5 levels, 4 heads, 1 head
    ("col_P4_model", dict(seed=109, N=2, Lq=n_px(P4), shapes=P4, model_like=True)),
    ("col_tiny_top_level", dict(seed=110, N=1, Lq=n_px([(1, 2), (3, 5), (50, 70)]), shapes=[(1, 2), (3, 5), (50, 70)],
                                model_like=True)),
    ("col_L5", dict(seed=111, N=1, Lq=n_px([(4, 4), (8, 8), (16, 16), (32, 32), (64, 64)]),
                    shapes=[(4, 4), (8, 8), (16, 16), (32, 32), (64, 64)], model_like=True, px_sigma=1.0)),
    ("col_levels_fine_to_coarse", dict(seed=112, N=1, Lq=n_px(P1), shapes=P1[::-1], model_like=True)),
    ("col_M4", dict(seed=113, N=2, M=4, Lq=n_px([(16, 20), (40, 50), (9, 9)]), shapes=[(16, 20), (40, 50), (9, 9)],
                    model_like=True)),
    ("col_M1_wide_strip", dict(seed=114, N=3, M=1, Lq=n_px([(2, 90), (4, 180), (8, 360)]),
                               shapes=[(2, 90), (4, 180), (8, 360)], model_like=True)),
    # one-row maps 60 000 pixels long: 3 750 columns along x -- the cell tables would take most of the pool, so the launch
    # runs on FLAT columns (256 consecutive queries of one level)
    ("col_flat_columns_long_strip", dict(seed=115, N=1, M=2, Lq=n_px([(1, 15000), (1, 30000), (1, 60000)]),
                                         shapes=[(1, 15000), (1, 30000), (1, 60000)], model_like=True, px_sigma=1.5, atol=2e-3)),
    # a level wider than the 16-bit box corners can express (65 531): every level through the global-memory path
    ("col_wider_than_the_box_corners", dict(seed=116, N=1, M=2, Lq=n_px([(1, 17500), (1, 35000), (1, 70000)]),
                                            shapes=[(1, 17500), (1, 35000), (1, 70000)], model_like=True, px_sigma=1.5, atol=2e-3)),
]


@pytest.mark.parametrize("cid,kw", COL_CASES, ids=[c[0] for c in COL_CASES])
def test_column_kernel_vs_oracle(MSDA, lib, cid, kw):
    kw = dict(dict(M=8, D=16, P=4), **kw)
    # (atol override: on a map 60 000 pixels wide one ulp of a fp32 pixel coordinate is 2^-8 px, and `loc * W - 0.5` rounds
    # once as an FMA -- here and under nvcc's default contraction in the reference -- but twice in the oracle's plain C)
    atol = kw.pop("atol", 1e-4)
    c = make_case(dtype=np.float32, **kw)
    want = orc.forward(c["value"], c["shapes"], c["starts"], c["loc"], c["attn"])
    with force(lib, K_COL):
        got = run_fwd(MSDA, c)
        assert lib.pct_msda_last_kernel() == K_COL
    np.testing.assert_allclose(got, want, rtol=0, atol=atol)
    if atol > 1e-4:                                  # ... and almost every element still agrees to 1e-4
        assert float(np.mean(np.abs(got - want) <= 1e-4)) > 0.9999


def test_column_kernel_inf_in_a_window_does_not_leak(MSDA, lib):
    """A non-finite texel inside a staged window reaches only the outputs whose samples really read it; gated-out
    samples read the zero pixels."""
    c = make_case(seed=121, N=1, M=8, D=16, Lq=n_px(P1), P=4, shapes=P1, model_like=True, px_sigma=1.0)
    c["value"][0, 100, 3, :] = np.inf
    c["value"][0, 3000, 5, :] = np.nan
    c["loc"][0, 2000:2050] += 3.0          # far outside the map: gated out
    want = orc.forward(c["value"], c["shapes"], c["starts"], c["loc"], c["attn"])
    with force(lib, K_COL):
        got = run_fwd(MSDA, c)
        assert lib.pct_msda_last_kernel() == K_COL
    np.testing.assert_array_equal(np.isfinite(got), np.isfinite(want))
    fin = np.isfinite(want)
    np.testing.assert_allclose(got[fin], want[fin], rtol=0, atol=1e-4)
    assert np.all(got[0, 2000:2050] == 0)


def _fused_case(shapes, P, N, shared_ref, seed=5, off_scale=3.0):
    rng = np.random.RandomState(seed)
    sh = np.asarray(shapes, dtype=np.int64)
    L, M, D = len(shapes), 8, 16
    S = int((sh[:, 0] * sh[:, 1]).sum())
    value = rng.standard_normal((N, S, M, D)).astype(np.float32)
    offsets = (rng.standard_normal((N, S, M, L, P, 2)) * off_scale).astype(np.float32)
    logits = (rng.standard_normal((N, S, M, L * P)) * 2).astype(np.float32)
    ref = rng.random_sample((1 if shared_ref else N, S, L, 2)).astype(np.float32)
    return sh, value, offsets, logits, ref


def _fused_check(MSDA, lib, shapes, P, N, shared_ref, kernel, expect, off_scale=3.0, pixel_ref=False):
    """fused(value, ref, offsets, logits) against the oracle on the module's own location / softmax math
    (ops/modules/ms_deform_attn.py:100-109) evaluated on the host in float64 -> float32."""
    sh, value, offsets, logits, ref = _fused_case(shapes, P, N, shared_ref, off_scale=off_scale)
    L = len(shapes)
    if pixel_ref:        # the encoder's reference points: every query's own pixel centre on every level
        from msda_cases import pixel_centres
        ref = np.broadcast_to(pixel_centres(sh)[None, :, None, :], (ref.shape[0], value.shape[1], L, 2)).astype(np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    ref_t = t(ref).expand(N, -1, -1, -1) if shared_ref else t(ref)
    with force(lib, kernel):
        got = MSDA.ms_deform_attn_fused_forward(t(value), t(sh), t(starts_of(sh)), ref_t, t(offsets), t(logits))
        ran = lib.pct_msda_last_kernel()
    assert ran == expect, "kernel %d ran, expected %d" % (ran, expect)
    norm = np.stack([sh[:, 1], sh[:, 0]], -1).astype(np.float32)
    loc = (np.broadcast_to(ref, (N,) + ref.shape[1:])[:, :, None, :, None, :]
           + offsets / norm[None, None, None, :, None, :]).astype(np.float32)
    lg = logits.astype(np.float64)
    w = np.exp(lg - lg.max(-1, keepdims=True))
    w = (w / w.sum(-1, keepdims=True)).astype(np.float32).reshape(N, -1, 8, L, P)
    want = orc.forward(value, sh, starts_of(sh), loc, w)
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=0, atol=1e-4)


@pytest.mark.parametrize("shapes,N,shared_ref,off_scale,pixel_ref", [
    (P2, 2, True, 3.0, False),          # random reference points: boxes as wide as the levels
    (P2, 1, True, 2.0, True),           # the encoder's own reference points, offsets ~ N(0, 2 px)
    (P4, 2, False, 3.0, False),
    (P1, 2, False, 1.0, True),
    ([(5, 7), (9, 9), (20, 31)], 3, False, 3.0, False),
])
def test_fused_front_end_on_the_column_kernel(MSDA, lib, shapes, N, shared_ref, off_scale, pixel_ref):
    _fused_check(MSDA, lib, shapes, 4, N, shared_ref, K_COL, K_COL, off_scale, pixel_ref)


@pytest.mark.parametrize("shapes,P,N,shared_ref", [
    (P2, 4, 2, True), (P4, 4, 2, False), ([(4, 4), (8, 8), (16, 16), (32, 32), (64, 64)], 8, 1, True),
])
def test_fused_front_end_on_the_windowed_kernel(MSDA, lib, shapes, P, N, shared_ref):
    """The fused windowed instantiations (round 1's bench kernel among them), forced and checked against the oracle."""
    _fused_check(MSDA, lib, shapes, P, N, shared_ref, K_WIN, K_WIN)


def test_fused_front_end_auto_route_at_model_batch(MSDA, lib):
    """P2 at N = 4 under `auto` (the column kernel from N = 1 on at this shape); the oracle does it in seconds."""
    _fused_check(MSDA, lib, P2, 4, 4, True, -1, K_COL, 2.0, True)


# ---------------------------------------------------------------- which kernel `auto` launches per BASELINE configuration
def _launch(MSDA, lib, shapes, N, P, dtype, fused=False):
    sh = np.asarray(shapes, dtype=np.int64)
    S = n_px(shapes)
    L = len(shapes)
    g = torch.Generator(device="cuda").manual_seed(0)
    v = torch.randn(N, S, 8, 16, device="cuda", generator=g).to(dtype)
    if fused:
        ref = torch.rand(1, S, L, 2, device="cuda", generator=g).expand(N, -1, -1, -1)
        MSDA.ms_deform_attn_fused_forward(v, dev(sh), dev(starts_of(sh)), ref,
                                          torch.randn(N, S, 8, L, P, 2, device="cuda", generator=g),
                                          torch.randn(N, S, 8, L * P, device="cuda", generator=g))
    else:
        MSDA.ms_deform_attn_forward(v, dev(sh), dev(starts_of(sh)), torch.rand(N, S, 8, L, P, 2, device="cuda", generator=g),
                                    torch.rand(N, S, 8, L, P, device="cuda", generator=g), 64)
    torch.cuda.synchronize()
    return lib.pct_msda_last_kernel()


@pytest.mark.parametrize("name,shapes,N,P,dtype,fused,expect", [
    # BASELINE.json configs[0]: 256^2 tile, batch 1 (S = 1344): too small for a persistent grid
    ("cfg1_256_b1", [(8, 8), (16, 16), (32, 32)], 1, 4, torch.float32, True, K_DPP),
    # configs[1]: 512^2, 4 levels; per-GPU batch 1, 8 and the bench's 128
    ("cfg2_512_b1", P2, 1, 4, torch.float32, True, K_COL),
    ("cfg2_512_b8", P2, 8, 4, torch.float32, True, K_COL),
    ("cfg2_512_b128_plain", P2, 128, 4, torch.float32, False, K_COL),
    ("cfg2_512_b128_fused", P2, 128, 4, torch.float32, True, K_COL),
    # configs[2]: CVPPP training crop 448^2, 3 levels, 2 images per GPU (S = 4116)
    ("cfg3_cvppp_b2", [(14, 14), (28, 28), (56, 56)], 2, 4, torch.float32, False, K_DPP),
    # configs[3]: BBBC 520x696 test tiles, 3 levels; batch 8
    ("cfg4_bbbc_b8", P4, 8, 4, torch.float32, True, K_COL),
    # configs[4]: 1024^2, 5 levels, 8 points, fp16 (S = 87296)
    ("cfg5_1024_b1_f16", P3, 1, 8, torch.float16, False, K_COL),
    ("cfg5_1024_b1_bf16", P3, 1, 8, torch.bfloat16, False, K_COL),
    ("cfg2_512_b8_bf16_value", P2, 8, 4, torch.bfloat16, False, K_COL),
])
def test_auto_kernel_choice_per_baseline_config(MSDA, lib, name, shapes, N, P, dtype, fused, expect):
    """Pins `auto`'s routing, so that a threshold change cannot silently strand a kernel without oracle coverage: every
    (kernel, dtype, P, fused) combination listed here has an oracle test on that very kernel in this file or in
    test_msda_gpu.py."""
    lib.pct_msda_set_kernel_choice(-1)
    assert _launch(MSDA, lib, shapes, N, P, dtype, fused) == expect


# ---------------------------------------------------------------- config 5 on its default route (quad-owner, 16-bit, P = 8)
@pytest.mark.parametrize("tdt,eps", [(torch.float16, 2.0 ** -11), (torch.bfloat16, 2.0 ** -8)])
@pytest.mark.parametrize("N,shapes,kw", [
    (2, [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64)], dict(model_like=True)),
    (1, [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64)], dict(lo=-0.1, hi=1.1)),
    (1, P3, dict(model_like=True)),                       # BASELINE.json configs[4] at its true size: S = 87 296
])
def test_config5_16bit_P8_L5_default_route_vs_oracle(MSDA, lib, tdt, eps, N, shapes, kw):
    """fp16 / bf16 value, 5 levels, 8 points through whatever `auto` launches (small: the quad-owner kernel; configs[4]
    at its true size: the 16-bit column kernel): fp32 oracle on the same 16-bit-rounded value, error within output
    rounding (new capability: the reference op is fp32 / fp64 only, cu:69,139)."""
    S = n_px(shapes)
    c = make_case(seed=131, N=N, M=8, D=16, Lq=S, P=8, shapes=shapes, **kw)
    v16 = torch.from_numpy(c["value"]).to(tdt)
    want = orc.forward(v16.float().numpy(), c["shapes"], c["starts"], c["loc"], c["attn"])
    lib.pct_msda_set_kernel_choice(-1)
    got = MSDA.ms_deform_attn_forward(v16.cuda(), dev(c["shapes"]), dev(c["starts"]), dev(c["loc"]), dev(c["attn"]), 64)
    assert lib.pct_msda_last_kernel() == (K_COL if N * S * 8 * 2 >= 160000 else K_DPP) and got.dtype == tdt
    err = np.abs(got.float().cpu().numpy() - want)
    slack = 2e-3 if max(w for _, w in shapes) > 20000 else 1e-5          # (fp32 coordinate resolution on very wide maps, see above)
    assert np.all(err <= eps * np.abs(want) + slack), float(err.max())


def test_config5_fullsize_properties(MSDA, lib):
    """configs[4] at its true size, fp16: constant value -> constant output; linear in value; pixel-centre locations
    are an exact gather (the three size-independent properties test_msda_gpu.py checks at the north-star size)."""
    sh = np.asarray(P3, dtype=np.int64)
    S, L, P, M, D, N = n_px(P3), 5, 8, 8, 16, 1
    g = torch.Generator(device="cuda").manual_seed(3)
    shd, std = dev(sh), dev(starts_of(sh))
    lo = torch.tensor([[0.5 / w, 0.5 / h] for h, w in P3], device="cuda").view(1, 1, 1, L, 1, 2)
    loc = lo + torch.rand(N, S, M, L, P, 2, device="cuda", generator=g) * (1 - 2 * lo)     # strictly interior
    a = torch.rand(N, S, M, L, P, device="cuda", generator=g) + 1e-3
    a = a / a.sum((-1, -2), keepdim=True)
    v = torch.full((N, S, M, D), 1.75, device="cuda", dtype=torch.float16)
    out = MSDA.ms_deform_attn_forward(v, shd, std, loc, a, 64)
    assert float((out.float() - 1.75).abs().max()) < 2e-3
    v1 = torch.randn(N, S, M, D, device="cuda", generator=g).half()
    v2 = torch.randn(N, S, M, D, device="cuda", generator=g).half()
    f = lambda x: MSDA.ms_deform_attn_forward(x, shd, std, loc, a, 64).float()
    o1, o2, o12 = f(v1), f(v2), f((v1.float() * 0.5 + v2.float() * 0.25).half())      # exact in fp16 up to rounding
    assert float((o12 - (0.5 * o1 + 0.25 * o2)).abs().max()) < 5e-3
    # exact gather
    h, w = 256, 256
    px = torch.randint(0, w, (N, S, M), device="cuda", generator=g)
    py = torch.randint(0, h, (N, S, M), device="cuda", generator=g)
    loc1 = torch.full((N, S, M, L, P, 2), 0.5, device="cuda")
    loc1[:, :, :, 4, 0, 0] = (px + 0.5) / w
    loc1[:, :, :, 4, 0, 1] = (py + 0.5) / h
    a1 = torch.zeros(N, S, M, L, P, device="cuda")
    a1[:, :, :, 4, 0] = 1.0
    out = MSDA.ms_deform_attn_forward(v1, shd, std, loc1, a1, 64).view(N, S, M, D)
    idx = int(starts_of(sh)[4]) + py * w + px
    want = torch.gather(v1, 1, idx[..., None].expand(N, S, M, D))
    assert torch.equal(out, want)
    assert lib.pct_msda_last_kernel() == K_COL


# ---------------------------------------------------------------- the module with a padding mask, on the device
@pytest.mark.parametrize("name,kw", [("module_L3_d128", dict(d_model=128, n_levels=3, n_heads=8, n_points=4)),
                                     ("module_L2_d64_mask", dict(d_model=64, n_levels=2, n_heads=4, n_points=2))])
def test_module_on_device_matches_reference_module_golden(MSDA, golden, name, kw):
    """MSDeformAttn on the GPU against the fixture produced by the reference's own module
    (ops/modules/ms_deform_attn.py:82-125), including input_padding_mask (value.masked_fill, :100-101), D = 16 with
    2 points per level (generic kernel) -- forward-only and with autograd."""
    from pctrans_amd.pixel_decoder.ops.modules import MSDeformAttn
    g = golden(name)
    m = MSDeformAttn(**kw)
    m.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd.")}, strict=True)
    m = m.cuda().eval()
    args = [torch.from_numpy(g[k]).cuda() for k in ("query", "ref", "src", "shapes", "starts")]
    mask = torch.from_numpy(g["mask"]).cuda() if "mask" in g else None
    with torch.no_grad():
        out = m(*args, mask)
    np.testing.assert_allclose(out.cpu().numpy(), g["out"], rtol=0, atol=1e-4)
    out2 = m(*args, mask)
    np.testing.assert_allclose(out2.detach().cpu().numpy(), g["out"], rtol=0, atol=1e-4)
    out2.square().sum().backward()
    assert torch.isfinite(m.value_proj.weight.grad).all() and float(m.value_proj.weight.grad.abs().max()) > 0


# ---------------------------------------------------------------- batches whose value tensor passes 2 GiB
def test_batch_256_is_chunked_inside_the_library_not_rerouted(MSDA, lib):
    """N = 256 at the north-star shape: value is 2.85 GB, past the 32-bit byte offsets of one launch.  The C ABI sends it
    out in chunks of images (as the reference does with im2col_step, cu:66-80) on the same kernel -- it used to drop to
    the unfused path at a third of the speed.  Checked: same kernel, every image equal to what a 2-image call gives."""
    sh = np.asarray(P2, dtype=np.int64)
    S, L, P, M, D, N = n_px(P2), 4, 4, 8, 16, 256
    g = torch.Generator(device="cuda").manual_seed(7)
    shd, std = dev(sh), dev(starts_of(sh))
    # two distinct images, repeated 128 times: the expected output is known from a 2-image call
    v2 = torch.randn(2, S, M, D, device="cuda", generator=g)
    off2 = torch.randn(2, S, M, L, P, 2, device="cuda", generator=g) * 2.0
    lg2 = torch.randn(2, S, M, L * P, device="cuda", generator=g)
    from msda_cases import pixel_centres
    ref = torch.from_numpy(np.broadcast_to(pixel_centres(sh)[None, :, None, :], (1, S, L, 2)).astype(np.float32).copy()).cuda()
    want2 = MSDA.ms_deform_attn_fused_forward(v2, shd, std, ref.expand(2, -1, -1, -1), off2, lg2)
    v = v2.repeat(N // 2, 1, 1, 1)
    off = off2.repeat(N // 2, 1, 1, 1, 1, 1)
    lg = lg2.repeat(N // 2, 1, 1, 1)
    assert v.numel() * 4 > 2 ** 31
    lib.pct_msda_set_kernel_choice(-1)
    got = MSDA.ms_deform_attn_fused_forward(v, shd, std, ref.expand(N, -1, -1, -1), off, lg)
    assert lib.pct_msda_last_kernel() == K_COL
    assert float((got.view(N // 2, 2, S, M * D) - want2[None]).abs().max()) <= 1e-5
    # the plain op too
    norm = torch.stack([shd[:, 1], shd[:, 0]], -1).float()
    loc2 = (ref[:, :, None, :, None, :] + off2 / norm[None, None, None, :, None, :]).contiguous()
    w2 = torch.softmax(lg2, -1).view(2, S, M, L, P).contiguous()
    want2p = MSDA.ms_deform_attn_forward(v2, shd, std, loc2, w2, 64)
    del off, lg
    gotp = MSDA.ms_deform_attn_forward(v, shd, std, loc2.repeat(N // 2, 1, 1, 1, 1, 1), w2.repeat(N // 2, 1, 1, 1, 1), 128)
    assert float((gotp.view(N // 2, 2, S, M * D) - want2p[None]).abs().max()) <= 1e-5


# ---------------------------------------------------------------- the 16-bit column kernel (msda_forward_col16.hip), forced
COL16_CASES = [
    # (id, P, shapes, make_case kwargs)
    ("c16_P8_L5_model", 8, [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64)], dict(N=2, model_like=True)),
    ("c16_P8_L5_edges", 8, [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64)], dict(N=1, lo=-0.2, hi=1.2)),
    ("c16_P8_L4_init_like", 8, P2, dict(N=1, init_like=True)),
    ("c16_P8_L3_nonpow2_sigma6", 8, P4, dict(N=2, model_like=True, px_sigma=6.0)),
    ("c16_P4_L4_model", 4, P2, dict(N=1, model_like=True)),
    ("c16_P4_L4_init_like", 4, P2, dict(N=2, init_like=True)),
    ("c16_P4_L3_uniform", 4, P1, dict(N=2)),
    ("c16_P4_L5_model", 4, [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64)], dict(N=2, model_like=True, px_sigma=1.0)),
    ("c16_P8_M3_heads", 8, [(9, 12), (18, 24), (36, 48)], dict(N=2, M=3, model_like=True)),
    ("c16_P4_flat_columns_long_strip", 4, [(1, 15000), (1, 30000), (1, 60000)], dict(N=1, M=2, model_like=True, px_sigma=1.5)),
]


@pytest.mark.parametrize("tdt,eps", [(torch.float16, 2.0 ** -11), (torch.bfloat16, 2.0 ** -8)])
@pytest.mark.parametrize("cid,P,shapes,kw", COL16_CASES, ids=[c[0] for c in COL16_CASES])
def test_column_kernel_16bit_vs_oracle(MSDA, lib, cid, P, shapes, kw, tdt, eps):
    kw = dict(dict(M=8), **kw)
    c = make_case(seed=141, D=16, Lq=n_px(shapes), P=P, shapes=shapes, **kw)
    v16 = torch.from_numpy(c["value"]).to(tdt)
    want = orc.forward(v16.float().numpy(), c["shapes"], c["starts"], c["loc"], c["attn"])
    with force(lib, K_COL):
        got = MSDA.ms_deform_attn_forward(v16.cuda(), dev(c["shapes"]), dev(c["starts"]), dev(c["loc"]), dev(c["attn"]), 64)
        assert lib.pct_msda_last_kernel() == K_COL and got.dtype == tdt
    err = np.abs(got.float().cpu().numpy() - want)
    slack = 2e-3 if max(w for _, w in shapes) > 20000 else 1e-5          # (fp32 coordinate resolution on very wide maps, see above)
    assert np.all(err <= eps * np.abs(want) + slack), float(err.max())
