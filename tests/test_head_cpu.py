"""CPU: host logic of the head modules -- golden vectors for the importable reference pieces (MSDeformAttn module
math, sine position encoding), re-statement equivalences for the pieces the reference cannot import (SURVEY.md 8c:
parity unpinned by the reference's own tests), and the state-dict key contract of SURVEY.md 8b.

CPU tensors only reach MSDeformAttn through the explicit `allow_cpu_reference` opt-in (dense torch formulation);
the product path for device tensors is the HIP kernel and is covered by the -m gpu tests.
"""
import re

import numpy as np
import pytest
import torch
from torch.nn import functional as F

from pctrans_amd.config import get_cfg, resnet_output_shape
from pctrans_amd.meta_arch.mask_former_head import MaskFormerHead
from pctrans_amd.pixel_decoder.ops.functions import ms_deform_attn_core_pytorch
from pctrans_amd.pixel_decoder.ops.modules import MSDeformAttn
from pctrans_amd.pixel_decoder.ops.modules import ms_deform_attn as msda_mod
from pctrans_amd.transformer_decoder import mask2former_transformer_decoder as dec
from pctrans_amd.transformer_decoder.attention import MultiheadAttention, attention_core
from pctrans_amd.transformer_decoder.position_encoding import PositionEmbeddingSine


@pytest.fixture()
def cpu_reference():
    prev = msda_mod.allow_cpu_reference(True)
    yield
    msda_mod.allow_cpu_reference(prev)


def test_position_encoding_matches_reference(golden):
    g = golden("position_encoding_sine")
    pe = PositionEmbeddingSine(64, normalize=True)
    for k, v in g.items():
        h, w = map(int, k[3:].split("x"))
        out = pe(torch.zeros(2, 3, h, w))
        np.testing.assert_allclose(out.numpy(), v, rtol=0, atol=1e-6)
        out2 = pe(torch.zeros(2, 3, h, w), mask=torch.zeros(2, h, w, dtype=torch.bool))   # uncached branch
        np.testing.assert_allclose(out2.numpy(), v, rtol=0, atol=1e-6)


@pytest.mark.parametrize("name,kw", [("module_L3_d128", dict(d_model=128, n_levels=3, n_heads=8, n_points=4)),
                                     ("module_L2_d64_mask", dict(d_model=64, n_levels=2, n_heads=4, n_points=2))])
def test_msdeformattn_module_matches_reference(golden, cpu_reference, name, kw):
    g = golden(name)
    m = MSDeformAttn(**kw)
    m.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd.")}, strict=True)
    args = [torch.from_numpy(g[k]) for k in ("query", "ref", "src", "shapes", "starts")]
    mask = torch.from_numpy(g["mask"]) if "mask" in g else None
    with torch.no_grad():
        out = m(*args, mask)
    np.testing.assert_allclose(out.numpy(), g["out"], rtol=0, atol=2e-6)


def test_msdeformattn_module_cpu_raises_without_opt_in():
    m = MSDeformAttn(32, 1, 4, 2)
    shapes = torch.tensor([[2, 2]])
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        m(torch.zeros(1, 4, 32), torch.zeros(1, 4, 1, 2), torch.zeros(1, 4, 32), shapes, torch.tensor([0]))


def test_reset_parameters_recipe():
    """ops/modules/ms_deform_attn.py:66-80: zero offset weights, head-directional bias scaled by point index."""
    m = MSDeformAttn(128, 3, 8, 4).requires_grad_(False)
    assert float(m.sampling_offsets.weight.abs().max()) == 0 and float(m.attention_weights.weight.abs().max()) == 0
    b = m.sampling_offsets.bias.detach().view(8, 3, 4, 2)
    np.testing.assert_allclose(b[0, :, :, 0].numpy(), np.tile([1, 2, 3, 4], (3, 1)), atol=1e-6)   # head 0: +x
    np.testing.assert_allclose(b[0, :, :, 1].numpy(), 0, atol=1e-6)
    np.testing.assert_allclose(b[2, 1, :, 1].numpy(), [1, 2, 3, 4], atol=1e-6)                     # head 2: +y
    assert float(b.abs().amax(-1).max()) == 4.0


def test_core_pytorch_matches_reference_fixture(golden):
    g = golden("fwd_nonpow2_edges_f32")
    out = ms_deform_attn_core_pytorch(*[torch.from_numpy(g[k]) for k in ("value", "shapes", "loc", "attn")])
    np.testing.assert_allclose(out.numpy(), g["out"], rtol=0, atol=1e-6)


# ---- attention: independent explicit formula (attention.py:271-387 has no reference test) ---------------------
def _explicit_attention(q, k, v, heads, mask):
    L, N, E = q.shape
    S, Ev = k.shape[0], v.shape[2]
    hd, vd = E // heads, Ev // heads
    out = torch.zeros(L, N, Ev, dtype=q.dtype)
    for n in range(N):
        for h in range(heads):
            qq = q[:, n, h * hd:(h + 1) * hd] * hd ** -0.5
            kk = k[:, n, h * hd:(h + 1) * hd]
            s = qq @ kk.t()
            if mask is not None:
                s = s.masked_fill(mask[n * heads + h], float("-inf"))
            out[:, n, h * vd:(h + 1) * vd] = torch.softmax(s, -1) @ v[:, n, h * vd:(h + 1) * vd]
    return out


def test_attention_core_against_explicit_formula():
    torch.manual_seed(0)
    L, S, N, heads = 7, 33, 2, 8
    q, k, v = torch.randn(L, N, 256, dtype=torch.float64), torch.randn(S, N, 256, dtype=torch.float64), \
        torch.randn(S, N, 128, dtype=torch.float64)
    mask = torch.rand(N * heads, L, S) < 0.6
    mask[:, :, 0] = False                      # no fully masked rows
    want = _explicit_attention(q, k, v, heads, mask)
    got, w = attention_core(q, k, v, heads, attn_mask=mask, need_weights=True)
    np.testing.assert_allclose(got.numpy(), want.numpy(), atol=1e-12)
    assert w.shape == (N, L, S) and abs(float(w.sum(-1).mean()) - 1) < 1e-12
    # broadcast-over-heads mask form used by the decoder == the per-head repeated form of the reference
    m1 = torch.rand(N, 1, L, S) < 0.5
    m1[..., 0] = False
    a, _ = attention_core(q, k, v, heads, attn_mask=m1)
    b, _ = attention_core(q, k, v, heads, attn_mask=m1.repeat(1, heads, 1, 1).flatten(0, 1))
    np.testing.assert_allclose(a.numpy(), b.numpy(), atol=0)
    # all-False mask == no mask
    c, _ = attention_core(q, k, v, heads, attn_mask=torch.zeros(L, S, dtype=torch.bool))
    d, _ = attention_core(q, k, v, heads)
    np.testing.assert_allclose(c.numpy(), d.numpy(), atol=0)


def test_multihead_attention_module_keys_and_shapes():
    m = MultiheadAttention(256, 8, vdim=128)
    assert sorted(m.state_dict()) == ["out_proj.bias", "out_proj.weight"]
    assert m.out_proj.weight.shape == (128, 128) and float(m.out_proj.bias.detach().abs().max()) == 0
    out, w = m(torch.randn(5, 2, 256), torch.randn(9, 2, 256), torch.randn(9, 2, 128))
    assert out.shape == (5, 2, 128) and w is None


# ---- dynamic mask head: batched formulation == the reference's materialise-and-grouped-conv formulation --------
def _reference_formulation(decoder, mask_feats, reference_points, params, stride=4):
    """mask2former_transformer_decoder.py:647-697 restated literally: per-query input gather + relative coordinates
    concatenated on the channel axis, parse_dynamic_params, three grouped convs with N*Q groups."""
    N, C, H, W = mask_feats.shape
    Q = reference_points.shape[0]
    ref = reference_points.transpose(0, 1) * torch.tensor([W * stride, H * stride], dtype=mask_feats.dtype)
    prm = params.transpose(0, 1)
    loc = dec.compute_locations(H, W, stride, mask_feats.device)
    rel = (ref.reshape(N, Q, 1, 1, 2) - loc.reshape(1, 1, H, W, 2)).permute(0, 1, 4, 2, 3).flatten(-2, -1)
    feats = mask_feats[:, None].expand(N, Q, C, H, W).reshape(N, Q, C, H * W)
    x = torch.cat([rel.float().to(mask_feats.dtype), feats], dim=2).reshape(1, -1, H, W)   # `.float()` as :670
    weights, biases = dec.parse_dynamic_params(prm.flatten(0, 1), decoder.dynamic_mask_channels,
                                               decoder.weight_nums, decoder.bias_nums)
    return decoder.mask_heads_forward(x, weights, biases, N * Q).reshape(N, Q, H, W)


def _small_decoder(Q=6, rel_coord=True):
    torch.manual_seed(1)
    return dec.MultiScaleMaskedTransformerDecoder(
        128, True, hidden_dim=128, num_queries=Q, nheads=8, dim_feedforward=256, dec_layers=3, pre_norm=False,
        mask_dim=16, enforce_input_project=False, points_num=1, sem_loss_on=True, norm="BN", rel_coord=rel_coord)


@pytest.mark.parametrize("rel_coord", [True, False])
def test_dynamic_mask_head_batched_equals_reference_formulation(rel_coord):
    d = _small_decoder(rel_coord=rel_coord).double()
    N, Q, H, W = 2, 6, 9, 7
    torch.manual_seed(2)
    mf = torch.randn(N, 16, H, W, dtype=torch.float64)
    ref = torch.rand(Q, N, 2, dtype=torch.float64)
    prm = torch.randn(Q, N, d.num_gen_params, dtype=torch.float64)
    assert d.num_gen_params == (233 if rel_coord else 217)
    got = d.mask_heads_forward_batched(mf, ref.transpose(0, 1), prm.transpose(0, 1), 4, rel_coord)
    if rel_coord:
        want = _reference_formulation(d, mf, ref, prm)
    else:
        feats = mf[:, None].expand(N, Q, 16, H, W).reshape(1, -1, H, W)
        weights, biases = dec.parse_dynamic_params(prm.transpose(0, 1).flatten(0, 1), 8, d.weight_nums, d.bias_nums)
        want = d.mask_heads_forward(feats, weights, biases, N * Q).reshape(N, Q, H, W)
    np.testing.assert_allclose(got.numpy(), want.numpy(), atol=1e-10)

    logits, attn_mask = d.dynamic_mask_with_coords(mf, ref, prm, 4, rel_coord, (5, 4))
    assert logits.shape == (N, Q, 2 * H, 2 * W) and attn_mask.shape == (N, 1, Q, 20) and attn_mask.dtype == torch.bool
    a = F.interpolate(want, size=(5, 4), mode="bilinear", align_corners=False)
    np.testing.assert_array_equal(attn_mask[:, 0].numpy(), (a.sigmoid().flatten(2) < 0.5).numpy())


def test_dynamic_mask_head_fp32_gemm_form_equals_the_reference_formulation():
    """fp32 (the training path outside autocast): coordinate terms and biases folded into the three batched GEMMs -- values
    and the gradients of features, reference points and generated parameters against the reference formulation evaluated in
    fp64 on the same fp32 inputs, at a 512-pixel image size (where the separately rounded w * r and w * l products matter most),
    with bounds scaled by the pre-activation magnitudes."""
    d = _small_decoder()
    N, Q, H, W = 2, 6, 128, 128
    torch.manual_seed(3)
    mf = torch.randn(N, 16, H, W, requires_grad=True)
    ref = torch.rand(Q, N, 2, requires_grad=True)
    prm = (torch.randn(Q, N, d.num_gen_params) * 0.3).requires_grad_(True)
    got = d.mask_heads_forward_batched(mf, ref.transpose(0, 1), prm.transpose(0, 1), 4, True)
    go = torch.randn_like(got)
    g_got = torch.autograd.grad(got, (mf, ref, prm), go)
    d64 = _small_decoder().double()
    mf64, ref64, prm64 = (t.detach().double().requires_grad_(True) for t in (mf, ref, prm))
    want = _reference_formulation(d64, mf64, ref64, prm64)
    g_want = torch.autograd.grad(want, (mf64, ref64, prm64), go.double())
    # pre-activations reach |w| * 512 px * 8 inputs: fp32 rounding of such sums, three layers deep
    scale = float(want.detach().abs().max())
    assert float((got.detach().double() - want.detach()).abs().max()) <= 2e-5 * scale
    # gradients: a pre-activation within rounding of zero may take the other side of its ReLU in fp32 (a finite change of a few
    # elements, in ANY fp32 evaluation): all-element bound on the norm, element-wise bound on all but 1e-4 of the elements
    for a, b in zip(g_got, g_want):
        err = (a.double() - b).abs()
        assert float(err.norm()) <= 1e-3 * float(b.norm())
        if b.numel() >= 10000:
            assert float((err > 2e-4 * float(b.abs().max())).double().mean()) <= 1e-4
        else:                                    # (reference points: 24 sums over 16 384 pixels x 8 channels each, in fp32)
            assert float(err.max()) <= 2e-3 * float(b.abs().max())


def test_zero_rel_coord_weights_reduce_to_plain_conv_stack():
    """Invariant from SURVEY.md 8c: with the two relative-coordinate weights zeroed the head is a plain 3-layer MLP."""
    d = _small_decoder().double()
    N, Q, H, W = 1, 3, 4, 5
    mf = torch.randn(N, 16, H, W, dtype=torch.float64)
    prm = torch.randn(N, Q, d.num_gen_params, dtype=torch.float64)
    w0 = prm[..., :144].reshape(N, Q, 8, 18)
    w0[..., :2] = 0
    out = d.mask_heads_forward_batched(mf, torch.rand(N, Q, 2, dtype=torch.float64), prm, 4, True)
    w1, w2 = prm[..., 144:208].reshape(N, Q, 8, 8), prm[..., 208:216].reshape(N, Q, 1, 8)
    b0, b1, b2 = prm[..., 216:224], prm[..., 224:232], prm[..., 232:233]
    x = mf.reshape(N, 1, 16, H * W)
    x = torch.relu(w0[..., 2:] @ x + b0[..., None])
    x = torch.relu(w1 @ x + b1[..., None])
    x = (w2 @ x + b2[..., None]).reshape(N, Q, H, W)
    np.testing.assert_allclose(out.numpy(), x.numpy(), atol=1e-12)


def test_sineembed_and_inverse_sigmoid():
    p = torch.rand(5, 2, 2)
    e = dec.gen_sineembed_for_position(p)
    assert e.shape == (5, 2, 256)
    # first 128 = y, layout interleaves sin/cos over 64 frequency pairs, temperature 20
    t = 20 ** (2 * (torch.arange(128) // 2) / 128.0)
    np.testing.assert_allclose(e[..., 0].numpy(), torch.sin(p[..., 1] * 2 * np.pi / t[0]).numpy(), atol=1e-6)
    np.testing.assert_allclose(e[..., 129].numpy(), torch.cos(p[..., 0] * 2 * np.pi / t[1]).numpy(), atol=1e-6)
    x = torch.tensor([0.0, 1e-4, 0.3, 1.0])
    np.testing.assert_allclose(dec.inverse_sigmoid(x).numpy(),
                               np.log(np.array([1e-3, 1e-3, 0.3, 1.0]) / np.array([1.0, 0.9999, 0.7, 1e-3])),
                               rtol=1e-5)


# ---- the whole head on CPU: shapes, outputs dict, state-dict key contract ---------------------------------------
@pytest.fixture(scope="module")
def small_head():
    torch.manual_seed(0)
    cfg = get_cfg(num_queries=10, norm="BN", sem_norm="BN")
    shapes = resnet_output_shape(18)
    return MaskFormerHead(**MaskFormerHead.from_config(cfg, shapes)).eval(), shapes


def test_head_forward_cfg1_shapes(small_head, cpu_reference):
    head, shapes = small_head
    feats = {k: torch.randn(2, s.channels, 128 // s.stride, 160 // s.stride) for k, s in shapes.items()}
    with torch.no_grad():
        pred, mask_features = head(feats)
    # PCTrans resizes the res2 lateral DOWN to the finest encoder level (msdeformattn.py:347): stride-8 mask features
    assert mask_features.shape == (2, 128, 16, 20)
    assert pred["pred_masks"].shape == (2, 10, 32, 40)
    assert pred["reference_points"].shape == (2, 10, 2)
    assert len(pred["aux_outputs"]) == 9 and len(pred["aux_reference_points"]) == 8
    assert pred["sem_mask"].shape == (2, 1, 16, 20) and pred["indices_list"] == []
    assert all(torch.isfinite(a["pred_masks"]).all() for a in pred["aux_outputs"])


def test_state_dict_keys_match_the_reference_contract(small_head):
    head, _ = small_head
    keys = set(head.state_dict().keys())

    def has(prefix, names, suffixes=("weight", "bias")):
        for n in names:
            for s in suffixes:
                assert f"{prefix}{n}.{s}" in keys, f"{prefix}{n}.{s}"

    for i in range(3):
        has(f"pixel_decoder.input_proj.{i}.", ["0", "1"])
    assert "pixel_decoder.transformer.level_embed" in keys
    for i in range(6):
        p = f"pixel_decoder.transformer.encoder.layers.{i}."
        has(p + "self_attn.", ["sampling_offsets", "attention_weights", "value_proj", "output_proj"])
        has(p, ["norm1", "linear1", "linear2", "norm2"])
    has("pixel_decoder.", ["adapter_1", "layer_1"], suffixes=("weight",))
    has("pixel_decoder.", ["adapter_1.norm", "layer_1.norm"], suffixes=("weight", "bias", "running_mean"))
    for i in range(9):
        has(f"predictor.transformer_self_attention_layers.{i}.",
            ["sa_qcontent_proj", "sa_qpos_proj", "sa_kcontent_proj", "sa_kpos_proj", "sa_v_proj",
             "self_attn.out_proj", "norm1"])
        has(f"predictor.transformer_cross_attention_layers.{i}.",
            ["ca_qcontent_proj", "ca_qpos_proj", "ca_kcontent_proj", "ca_kpos_proj", "ca_v_proj",
             "ca_qpos_sine_proj", "cross_attn.out_proj", "norm2"])
        has(f"predictor.transformer_ffn_layers.{i}.", ["linear1", "linear2", "norm"])
    has("predictor.", ["decoder_norm", "mask_head", "logits", "ref_point_head.layers.0", "ref_point_head.layers.1",
                       "query_scale.layers.0", "query_scale.layers.1", "point_embed.layers.2",
                       "controller.layers.0", "controller.layers.2", "seg_head.0.1", "seg_head.1.1"])
    has("predictor.", ["seg_head.0.0", "seg_head.1.0"], suffixes=("weight",))
    for n in ("query_feat", "query_embed", "level_embed"):
        assert f"predictor.{n}.weight" in keys
    sd = head.state_dict()
    assert sd["predictor.controller.layers.2.weight"].shape == (233, 128)
    assert sd["predictor.transformer_cross_attention_layers.0.ca_qpos_sine_proj.weight"].shape == (128, 256)
    assert sd["predictor.query_scale.layers.1.weight"].shape == (256, 256)
    # nothing that is not a parameter/buffer of the reference modules
    assert not [k for k in keys if re.search(r"_cache|_geom", k)]


def test_four_level_north_star_geometry(cpu_reference):
    """res2..res5 into the encoder (north-star shape): no FPN stage, mask features at stride 4."""
    torch.manual_seed(0)
    cfg = get_cfg(num_queries=4, enc_in_features=("res2", "res3", "res4", "res5"), norm="BN", sem_norm="BN",
                  enc_layers=1, dec_layers=2)
    shapes = resnet_output_shape(18)
    head = MaskFormerHead(**MaskFormerHead.from_config(cfg, shapes)).eval()
    assert head.pixel_decoder.num_fpn_levels == 0 and head.pixel_decoder.transformer_num_feature_levels == 4
    feats = {k: torch.randn(1, s.channels, 64 // s.stride, 64 // s.stride) for k, s in shapes.items()}
    with torch.no_grad():
        pred, mf = head(feats)
    assert mf.shape == (1, 128, 16, 16) and pred["pred_masks"].shape == (1, 4, 32, 32)


# ---- query-contrast selection (training hook of the last decoder layer) ------------------------------------------
def _reference_style_selection(query, emb_dist, pos_indices):
    """mask2former_transformer_decoder.py:800-858 restated literally (list/set bookkeeping) for one batch."""
    out = []
    q = query.transpose(0, 1)
    Qn = q.shape[1]
    for b in range(q.shape[0]):
        pos_ids = pos_indices[b][0].tolist()
        rest_ids = list(set(range(Qn)) - set(pos_ids))
        rest = emb_dist[b][rest_ids][:, pos_ids]
        mdp = torch.argmax(rest, dim=1).tolist()
        mdp = torch.tensor([pos_ids[i] for i in mdp])
        for pid in pos_ids:
            cl = [rest_ids[i] for i in torch.where(mdp == pid)[0].tolist()]
            if not cl:
                continue
            neg = list(set(range(Qn)) - set(cl + [pid]))
            out.append((b, pid, sorted(cl), sorted(neg)))
    return out


def test_query_contrast_selection_matches_reference_bookkeeping():
    import random
    from pctrans_amd.transformer_decoder import query_contrast as qc
    torch.manual_seed(0)
    Q, N, C = 12, 2, 16
    output = torch.randn(Q, N, C)
    masks = torch.randn(N, Q, 6, 5)
    indices = [(torch.tensor([1, 4, 7]), torch.tensor([0, 1, 2])), (torch.tensor([0, 11]), torch.tensor([1, 0]))]
    random.seed(5)
    items_q, items_m = qc.query_contrast_items(output, masks, indices)
    qn = output.permute(1, 0, 2)
    emb = torch.stack([torch.cosine_similarity(qn[i].unsqueeze(1), qn[i].unsqueeze(0), dim=-1) for i in range(N)])
    want = _reference_style_selection(output, emb, indices)
    assert len(items_q) == len(items_m) == len(want) > 0
    for it_q, it_m, (b, pid, cl, neg) in zip(items_q, items_m, want):
        assert int(it_q["label"].sum()) == len(cl) and it_q["label"].numel() == len(cl) + len(neg)
        key = output[pid, b]
        np.testing.assert_allclose(it_q["contrast"][:len(cl), 0].numpy(), (output[cl, b] @ key).numpy(), atol=1e-5)
        np.testing.assert_allclose(it_q["contrast"][len(cl):, 0].numpy(), (output[neg, b] @ key).numpy(), atol=1e-5)
        n_s = len(neg) if len(cl) * 10 >= len(neg) else len(cl) * 10
        assert it_q["aux_consin"].shape == (len(cl) + n_s, 1) and float(it_q["aux_consin"].abs().max()) <= 1 + 1e-5
        d = qc.dice_for(masks[b])
        np.testing.assert_allclose(it_m["contrast"][:, 0].numpy(), torch.cat([d[pid][cl], d[pid][neg]]).numpy(),
                                   atol=1e-6)
    # dice_for: symmetric; exactly 1 for identical hard (0/1) masks
    d = qc.dice_for(masks[0])
    assert torch.allclose(d, d.t())
    hard = torch.where(torch.rand(3, 4, 4) < 0.5, 50.0, -50.0)
    assert torch.allclose(qc.dice_for(hard).diag(), torch.ones(3), atol=1e-6)
