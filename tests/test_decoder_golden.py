"""CPU: decoder-side math against vectors produced by the reference's OWN functions (tests/golden/make_golden_decoder.py
executes gen_sineembed_for_position, inverse_sigmoid, MLP, dynamic_mask_with_coords / mask_heads_forward /
parse_dynamic_params / compute_locations and dice_for straight from the reference file's AST).  fp32, tolerance 1e-5
relative to scale: the restatements reorder sums (batched matmul instead of N*Q grouped convolutions)."""
import numpy as np
import pytest
import torch

from pctrans_amd.transformer_decoder import mask2former_transformer_decoder as dec


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


@pytest.fixture()
def cpu_reference():
    from pctrans_amd.pixel_decoder.ops.modules import ms_deform_attn as msda_mod
    prev = msda_mod.allow_cpu_reference(True)
    yield
    msda_mod.allow_cpu_reference(prev)


def test_sineembed_and_inverse_sigmoid_match_the_reference_functions(golden):
    g = golden("dec_sineembed_inverse_sigmoid")
    for pts, emb, temp in ((g["pts2"], g["emb2"], 20), (g["pts4"], g["emb4"], 20), (g["pts2"], g["emb2_t10"], 10)):
        got = dec.gen_sineembed_for_position(_t(pts), temp)
        np.testing.assert_allclose(got.numpy(), emb, rtol=0, atol=2e-6)
    np.testing.assert_allclose(dec.inverse_sigmoid(_t(g["x"])).numpy(), g["inv"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(dec.inverse_sigmoid(_t(g["x"]), 1e-5).numpy(), g["inv_eps1e5"], rtol=1e-6, atol=1e-6)


def test_mlp_matches_the_reference_class_and_its_state_dict_keys(golden):
    g = golden("dec_mlp")
    mlp = dec.MLP(16, 24, 233, 3)
    sd = {k[3:]: _t(g[k]) for k in g if k.startswith("sd.")}
    assert sorted(sd) == sorted(mlp.state_dict())                  # same parameter names as the reference's MLP
    mlp.load_state_dict(sd)
    with torch.no_grad():
        np.testing.assert_allclose(mlp(_t(g["x"])).numpy(), g["y"], rtol=0, atol=1e-5)


@pytest.mark.parametrize("tag", ["rel", "norel", "rel_up"])
def test_dynamic_mask_head_batched_formulation_matches_the_reference_method(golden, tag):
    """`dynamic_mask_with_coords` (:647-697): per-query dynamic convs on [rel coords | mask features], x2 bilinear
    upsample, and the boolean attention mask at the next level's size (reference: repeated per head)."""
    g = golden("dec_dynamic_mask_head_" + tag)
    feats, refpts, params = _t(g["feats"]), _t(g["refpts"]), _t(g["params"])
    rel, stride, tgt, heads = bool(g["rel_coord"]), int(g["stride"]), tuple(int(v) for v in g["target"]), int(g["heads"])
    C = feats.shape[1]
    self = type("Bag", (), {})()
    self.dynamic_mask_channels, self.controller_layers = 8, 3
    self.weight_nums = [(C + 2 if rel else C) * 8, 64, 8]
    self.bias_nums = [8, 8, 1]
    self.mask_heads_forward_batched = lambda *a: dec.MultiScaleMaskedTransformerDecoder.mask_heads_forward_batched(self, *a)
    logits, amask = dec.MultiScaleMaskedTransformerDecoder.dynamic_mask_with_coords(
        self, feats, refpts, params, stride, rel, tgt)
    scale = max(1.0, float(np.abs(g["logits_x2"]).max()))
    np.testing.assert_allclose(logits.numpy(), g["logits_x2"], rtol=0, atol=1e-5 * scale)
    N, Q = feats.shape[0], refpts.shape[0]
    want_mask = g["attn_mask"].reshape(N, heads, Q, -1)
    assert (want_mask == want_mask[:, :1]).all()                   # the reference repeats one mask over the heads
    got_mask = amask.numpy()                                       # [N, 1, Q, hw], broadcast over heads
    # a logit within float noise of the sigmoid threshold may fall on either side
    flip = got_mask != want_mask[:, :1]
    assert flip.mean() < 1e-3
    np.testing.assert_allclose(dec.compute_locations(feats.shape[2], feats.shape[3], stride, "cpu").numpy(),
                               g["locations"], rtol=0, atol=0)


def test_decoder_dice_for_matches_the_reference_function(golden):
    from pctrans_amd.transformer_decoder import query_contrast as qc
    g = golden("dec_dice_for")
    np.testing.assert_allclose(qc.dice_for(_t(g["masks"])).numpy(), g["dice"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("tag", ["ca", "sa"])
def test_multihead_attention_matches_the_reference_class(golden, tag):
    """Projection-free MultiheadAttention (attention.py:57-387 of the reference, which does not import on torch >= 2):
    plain, boolean per-head mask, float mask + key padding mask; output and head-averaged weights."""
    from pctrans_amd.transformer_decoder.attention import MultiheadAttention
    g = golden("dec_attention_" + tag)
    q, k, v = _t(g["q"]), _t(g["k"]), _t(g["v"])
    mha = MultiheadAttention(q.shape[2], int(g["heads"]), dropout=0.0, vdim=v.shape[2]).eval()
    assert sorted(mha.state_dict()) == ["out_proj.bias", "out_proj.weight"]
    mha.load_state_dict({"out_proj.weight": _t(g["out_w"]), "out_proj.bias": _t(g["out_b"])})
    cases = ((dict(), "plain"), (dict(attn_mask=_t(g["bool_mask"])), "bool"),
             (dict(attn_mask=_t(g["float_mask"]), key_padding_mask=_t(g["key_padding_mask"])), "float_kpm"))
    with torch.no_grad():
        for kw, name in cases:
            out, w = mha(q, k, v, need_weights=True, **kw)
            np.testing.assert_allclose(out.numpy(), g["out_" + name], rtol=0, atol=2e-5, err_msg=name)
            np.testing.assert_allclose(w.numpy(), g["w_" + name], rtol=0, atol=2e-6, err_msg=name)


def test_decoder_layer_classes_match_the_reference_classes(golden):
    """SelfAttentionLayer / CrossAttentionLayer (first and later layers, with and without memory mask) / FFNLayer
    (dec.py:47-235): same state-dict keys, same outputs as the reference's classes run on their own attention."""
    g = golden("dec_layers")
    d, heads = g["tgt"].shape[2], int(g["heads"])
    tgt, qpos, mem, pos, qsine = (_t(g[k]) for k in ("tgt", "query_pos", "memory", "pos", "query_sine_embed"))
    mmask = _t(g["memory_mask"])
    layers = {"sa": dec.SelfAttentionLayer(d, heads).eval(), "ca": dec.CrossAttentionLayer(d, heads).eval(),
              "ffn": dec.FFNLayer(d, 2 * d).eval()}
    for name, m in layers.items():
        sd = {k[len("sd.%s." % name):]: _t(g[k]) for k in g if k.startswith("sd.%s." % name)}
        assert sorted(sd) == sorted(m.state_dict()), name          # the reference's parameter names
        m.load_state_dict(sd)
    sa, ca, ffn = layers["sa"], layers["ca"], layers["ffn"]
    with torch.no_grad():
        got = {"sa_out": sa(tgt, query_pos=qpos),
               "ca_first": ca(tgt, mem, memory_mask=mmask, pos=pos, query_pos=qpos, query_sine_embed=qsine, is_first=True),
               "ca_later": ca(tgt, mem, memory_mask=mmask, pos=pos, query_pos=qpos, query_sine_embed=qsine, is_first=False),
               "ca_nomask": ca(tgt, mem, pos=pos, query_pos=qpos, query_sine_embed=qsine, is_first=False),
               "ffn_out": ffn(tgt)}
    for k, v in got.items():
        np.testing.assert_allclose(v.numpy(), g[k], rtol=0, atol=2e-5, err_msg=k)


def test_query_contrast_selection_matches_the_reference_functions(golden):
    """select_pos_neg_query / select_pos_neg_mask (dec.py:800-901): same items in the same order, same 'contrast'
    scores and labels (the reference's randomly sub-sampled 'aux_*' entries are not part of the loss and not compared)."""
    from pctrans_amd.transformer_decoder import query_contrast as qc
    g = golden("dec_query_contrast")
    query, emb_dist, masks = _t(g["query"]), _t(g["emb_dist"]), _t(g["masks"])
    pos_indices = [(_t(g["pos_src_%d" % b]), _t(g["pos_tgt_%d" % b])) for b in range(query.shape[1])]
    items_q = qc.select_pos_neg_query(query, emb_dist, pos_indices)
    items_m = qc.select_pos_neg_mask(masks, emb_dist, pos_indices)
    assert len(items_q) == int(g["n_items_q"]) and len(items_m) == int(g["n_items_m"])
    # The reference lists the negatives in the iteration order of a Python set difference (a CPython hash-table detail:
    # {0, 8, 2} for 14 queries); the restatement lists them ascending.  The loss is a logsumexp over each group, so the
    # comparison is per group (positives first, then negatives) up to order.
    for prefix, items in (("q", items_q), ("m", items_m)):
        for i, it in enumerate(items):
            want_c, want_l = g["%s%d_contrast" % (prefix, i)].ravel(), g["%s%d_label" % (prefix, i)]
            got_c, got_l = it["contrast"].numpy().ravel(), it["label"].numpy()
            np.testing.assert_array_equal(got_l, want_l)
            for lab in (0, 1):
                np.testing.assert_allclose(np.sort(got_c[got_l == lab]), np.sort(want_c[want_l == lab]), rtol=1e-5,
                                           atol=1e-6)


def _encoder_from_fixture(g, device):
    from pctrans_amd.pixel_decoder.msdeformattn import MSDeformAttnTransformerEncoderOnly
    enc = MSDeformAttnTransformerEncoderOnly(d_model=32, nhead=4, num_encoder_layers=2, dim_feedforward=64, dropout=0.0,
                                             activation="relu", num_feature_levels=3, enc_n_points=4).eval()
    sd = {k[3:]: _t(g[k]) for k in g if k.startswith("sd.")}
    assert sorted(sd) == sorted(enc.state_dict())                  # the reference's parameter names
    enc.load_state_dict(sd)
    srcs = [_t(g["src%d" % i]).to(device) for i in range(3)]
    poss = [_t(g["pos%d" % i]).to(device) for i in range(3)]
    return enc.to(device), srcs, poss


def test_msdeform_encoder_matches_the_reference_classes(golden, cpu_reference):
    """MSDeformAttnTransformerEncoderOnly / EncoderLayer / Encoder (msdeformattn.py:23-162 of the reference, run on the
    reference's own MSDeformAttn module): level embedding, flattening order, reference points, two layers."""
    g = golden("dec_msdeform_encoder")
    enc, srcs, poss = _encoder_from_fixture(g, "cpu")
    with torch.no_grad():
        memory, shapes, starts = enc(srcs, poss)
    np.testing.assert_array_equal(shapes.numpy(), g["spatial_shapes"])
    np.testing.assert_array_equal(starts.numpy(), g["level_start_index"])
    np.testing.assert_allclose(memory.numpy(), g["memory"], rtol=0, atol=2e-5)


@pytest.mark.gpu
def test_msdeform_encoder_on_the_hip_kernels_matches_the_reference_classes(golden):
    g = golden("dec_msdeform_encoder")
    enc, srcs, poss = _encoder_from_fixture(g, "cuda")
    with torch.no_grad():
        memory, _, _ = enc(srcs, poss)
    np.testing.assert_allclose(memory.cpu().numpy(), g["memory"], rtol=0, atol=1e-4)


def _full_decoder(device):
    from golden_params import deterministic_fill
    d = dec.MultiScaleMaskedTransformerDecoder(
        128, True, hidden_dim=128, num_queries=6, nheads=8, dim_feedforward=256, dec_layers=3, pre_norm=False,
        mask_dim=16, enforce_input_project=False, points_num=1, sem_loss_on=False, norm="GN", rel_coord=True).eval()
    return deterministic_fill(d, 41).to(device)


def test_whole_decoder_forward_matches_the_reference_class(golden):
    """MultiScaleMaskedTransformerDecoder.forward in eval mode (dec.py:502-645: the layer loop, level cycling, reference
    point refinement, controller -> dynamic mask head, attention-mask hand-over) against the reference's own class run
    on the same name-derived parameters: same parameter names, same predictions, auxiliary outputs and points."""
    g = golden("dec_full_decoder")
    d = _full_decoder("cpu")
    assert sorted(d.state_dict()) == [str(n) for n in g["param_names"]]
    xs = [_t(g["x%d" % i]) for i in range(3)]
    with torch.no_grad():
        out = d(xs, None, _t(g["mask_features"]))
    scale = max(1.0, float(np.abs(g["pred_masks"]).max()))
    np.testing.assert_allclose(out["pred_masks"].numpy(), g["pred_masks"], rtol=0, atol=2e-4 * scale)
    np.testing.assert_allclose(out["reference_points"].numpy(), g["reference_points"], rtol=0, atol=2e-5)
    assert len(out["aux_outputs"]) == int(g["n_aux"])
    for i, a in enumerate(out["aux_outputs"]):
        np.testing.assert_allclose(a["pred_masks"].numpy(), g["aux%d_pred_masks" % i], rtol=0, atol=2e-4 * scale)
    for i, a in enumerate(out["aux_reference_points"]):
        np.testing.assert_allclose(a["reference_points"].numpy(), g["aux%d_reference_points" % i], rtol=0, atol=2e-5)


@pytest.mark.gpu
def test_whole_decoder_on_the_fused_kernels_matches_the_reference_class(golden):
    """Same, fp32 on the device: fused dynamic-mask-head kernel + torch attention (fp32 has no MFMA attention path)."""
    g = golden("dec_full_decoder")
    d = _full_decoder("cuda")
    xs = [_t(g["x%d" % i]).cuda() for i in range(3)]
    with torch.no_grad():
        out = d(xs, None, _t(g["mask_features"]).cuda())
    scale = max(1.0, float(np.abs(g["pred_masks"]).max()))
    np.testing.assert_allclose(out["pred_masks"].cpu().numpy(), g["pred_masks"], rtol=0, atol=5e-4 * scale)
    np.testing.assert_allclose(out["reference_points"].cpu().numpy(), g["reference_points"], rtol=0, atol=5e-5)


def test_instance_postprocessing_helpers_match_the_reference_functions(golden):
    """arch/maskformer.py of the reference: dice_for (:392-401), mask_post (:403-431; soft / hard merge / BBBC thresholds)
    and comput_mmi (:349-354)."""
    from pctrans_amd.arch import maskformer as mfm
    g = golden("arch_mask_post")
    inst = _t(g["inst_masks"])
    np.testing.assert_allclose(mfm.dice_for(inst).numpy(), g["dice"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(mfm.mask_post(inst, thres1=0.5, thres2=0.6, bd_flag=False).numpy(), g["post_soft"],
                               rtol=0, atol=1e-6)
    np.testing.assert_array_equal(mfm.mask_post(inst, thres1=0.5, thres2=0.6, bd_flag=True).numpy(), g["post_hard"])
    np.testing.assert_allclose(mfm.mask_post(inst, thres1=0.15, thres2=0.25).numpy(), g["post_bbbc"], rtol=0, atol=1e-6)
    for (a, b, c), want in zip(g["mmi_in"], g["mmi_out"]):
        got = float(mfm.comput_mmi(torch.tensor(float(a)), torch.tensor(float(b)), torch.tensor(float(c))))
        assert abs(got - float(want)) <= 1e-6 * max(1.0, abs(float(want)))


def test_loss_and_matcher_cost_definitions_match_the_reference_functions(golden):
    """dice_loss / sigmoid_ce_loss / calculate_uncertainty (maskformer_criterion.py:23-115) and batch_dice_loss /
    batch_sigmoid_ce_loss (matcher.py:15-62) of the reference."""
    from pctrans_amd.loss import maskformer_criterion as crit
    from pctrans_amd.loss import matcher
    g = golden("loss_functions")
    logits, tgt, tgt2, nm = _t(g["logits"]), _t(g["targets"]), _t(g["targets2"]), float(g["num_masks"])
    np.testing.assert_allclose(crit.dice_loss(logits, tgt, nm).numpy(), g["dice_loss"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(crit.sigmoid_ce_loss(logits, tgt, nm).numpy(), g["sigmoid_ce_loss"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(crit.calculate_uncertainty(logits[:, None, :]).numpy(), g["uncertainty"], rtol=0, atol=0)
    np.testing.assert_allclose(matcher.batch_dice_loss(logits, tgt2).numpy(), g["batch_dice"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(matcher.batch_sigmoid_ce_loss(logits, tgt2).numpy(), g["batch_ce"], rtol=1e-5, atol=1e-5)


# ---------------------------------------------------------------- pixel-decoder glue and the head ----
_PIX_CH = {"res2": 16, "res3": 24, "res4": 32, "res5": 40}
_PIX_STRIDE = {"res2": 4, "res3": 8, "res4": 16, "res5": 32}


def _pixel_decoder(device):
    """This package's MSDeformAttnPixelDecoder built with the reference's constructor keywords for the fixture's
    configuration (4 encoder levels, conv_dim 128, 8 heads, 2 encoder layers) and the name-derived parameters the
    generator gave the reference modules."""
    from golden_params import fill_pixel_decoder
    from pctrans_amd.layers import ShapeSpec
    from pctrans_amd.pixel_decoder.msdeformattn import MSDeformAttnPixelDecoder
    shapes = {k: ShapeSpec(channels=c, stride=_PIX_STRIDE[k]) for k, c in _PIX_CH.items()}
    pix = MSDeformAttnPixelDecoder(shapes, transformer_dropout=0.0, transformer_nheads=8, transformer_dim_feedforward=1024,
                                   transformer_enc_layers=2, conv_dim=128, mask_dim=16, norm="GN",
                                   transformer_in_features=["res2", "res3", "res4", "res5"], common_stride=4).eval()
    return fill_pixel_decoder(pix, 51).to(device)


def _check_pixel_decoder(g, out, atol):
    mask_features, enc_feat, multi = out
    scale = max(1.0, float(np.abs(g["mask_features"]).max()))
    np.testing.assert_allclose(mask_features.cpu().numpy(), g["mask_features"], rtol=0, atol=atol * scale)
    np.testing.assert_allclose(enc_feat.cpu().numpy(), g["transformer_encoder_features"], rtol=0, atol=atol * scale)
    assert len(multi) == 3
    for i, m_ in enumerate(multi):
        np.testing.assert_allclose(m_.cpu().numpy(), g["multi_scale_%d" % i], rtol=0, atol=atol * scale)


def test_pixel_decoder_forward_features_matches_the_reference_method(golden, cpu_reference):
    """MSDeformAttnPixelDecoder.forward_features (msdeformattn.py:314-360) -- input projections + GroupNorm, sine position
    embedding, level concat, the encoder, the split back into maps -- against the reference's own method run on the
    reference's modules (tests/golden/make_golden_decoder.py); same state-dict keys."""
    g = golden("dec_pixel_decoder_l4")
    pix = _pixel_decoder("cpu")
    assert sorted(pix.state_dict()) == [str(n) for n in g["param_names"]]
    feats = {k: _t(g["feat_" + k]) for k in _PIX_CH}
    with torch.no_grad():
        _check_pixel_decoder(g, pix.forward_features(feats), 1e-5)


@pytest.mark.gpu
def test_pixel_decoder_on_the_hip_kernels_matches_the_reference_method(golden):
    """Same on the device, fp32, forward-only: the 1x1 projections as GEMMs, pct_groupnorm_flatten_f32, the merged
    K = 128 projection kernels, the fused MSDeformAttn kernel and the fused output_proj / FFN + LayerNorm kernels, <= 1e-4
    (north_star)."""
    g = golden("dec_pixel_decoder_l4")
    pix = _pixel_decoder("cuda")
    feats = {k: _t(g["feat_" + k]).cuda() for k in _PIX_CH}
    from pctrans_amd import _lib
    with torch.no_grad():
        out = pix.forward_features(feats)
    assert _lib.lib().pct_msda_last_kernel() != 0               # the sampling really went through the HIP library
    _check_pixel_decoder(g, out, 1e-4)
    # and with autograd (unfused op + MSDeformAttnFunction): same numbers
    feats_g = {k: v.clone().requires_grad_() for k, v in feats.items()}
    _check_pixel_decoder(g, [o.detach() if torch.is_tensor(o) else [m_.detach() for m_ in o]
                             for o in pix.forward_features(feats_g)], 1e-4)


def _pixel_decoder3(device):
    """The SHIPPED geometry (configs/CVPPP/CVPPP-PCTrans.yaml:17-26): three encoder levels res3..res5 and one FPN level that
    brings res2 in (msdeformattn.py:255-290, 340-350)."""
    from golden_params import fill_pixel_decoder
    from pctrans_amd.layers import ShapeSpec
    from pctrans_amd.pixel_decoder.msdeformattn import MSDeformAttnPixelDecoder
    shapes = {k: ShapeSpec(channels=c, stride=_PIX_STRIDE[k]) for k, c in _PIX_CH.items()}
    pix = MSDeformAttnPixelDecoder(shapes, transformer_dropout=0.0, transformer_nheads=8, transformer_dim_feedforward=1024,
                                   transformer_enc_layers=2, conv_dim=128, mask_dim=16, norm="GN",
                                   transformer_in_features=["res3", "res4", "res5"], common_stride=4).eval()
    return fill_pixel_decoder(pix, 53).to(device)


def test_pixel_decoder_fpn_stage_matches_the_reference_method(golden, cpu_reference):
    """The FPN stage of forward_features (msdeformattn.py:340-350) -- what the shipped three-level yamls execute: lateral 1x1 of
    res2, bilinear resize DOWN to the finest encoder map (:347), sum, 3x3 output conv + norm + ReLU, outputs (out[-1], out[0],
    out[:3]) -- against the reference's own method body run on a bag whose two convolution wrappers the generator writes from
    detectron2's documented Conv2d semantics (the fixture pins the loop, the resize direction and the output order, not
    detectron2 itself); same state-dict keys (`adapter_1.*`, `layer_1.*`)."""
    g = golden("dec_pixel_decoder_l3_fpn")
    pix = _pixel_decoder3("cpu")
    assert pix.num_fpn_levels == 1
    assert sorted(pix.state_dict()) == [str(n) for n in g["param_names"]]
    feats = {k: _t(g["feat_" + k]) for k in _PIX_CH}
    with torch.no_grad():
        out = pix.forward_features(feats)
    assert tuple(out[0].shape) == tuple(g["mask_features"].shape) == (2, 128, 8, 10)      # stride 8: resized DOWN, not up
    _check_pixel_decoder(g, out, 1e-5)


@pytest.mark.gpu
def test_pixel_decoder_fpn_stage_on_the_hip_kernels_matches_the_reference_method(golden):
    """Same on the device, fp32, forward-only (the encoder on the HIP kernels, the FPN convolutions on MIOpen), <= 1e-4."""
    g = golden("dec_pixel_decoder_l3_fpn")
    pix = _pixel_decoder3("cuda")
    feats = {k: _t(g["feat_" + k]).cuda() for k in _PIX_CH}
    from pctrans_amd import _lib
    with torch.no_grad():
        out = pix.forward_features(feats)
    assert _lib.lib().pct_msda_last_kernel() != 0
    _check_pixel_decoder(g, out, 1e-4)
    feats_g = {k: v.clone().requires_grad_() for k, v in feats.items()}
    _check_pixel_decoder(g, [o.detach() if torch.is_tensor(o) else [m_.detach() for m_ in o]
                             for o in pix.forward_features(feats_g)], 1e-4)


def _head(device):
    from pctrans_amd.layers import ShapeSpec
    from pctrans_amd.meta_arch.mask_former_head import MaskFormerHead
    shapes = {k: ShapeSpec(channels=c, stride=_PIX_STRIDE[k]) for k, c in _PIX_CH.items()}
    return MaskFormerHead(shapes, num_classes=1, pixel_decoder=_pixel_decoder(device), loss_weight=1.0, ignore_value=-1,
                          transformer_predictor=_full_decoder(device), transformer_in_feature="multi_scale_pixel_decoder",
                          attn_mask_threshold=0.5).eval()


def _check_head(g, pred, mf, atol):
    scale = max(1.0, float(np.abs(g["pred_masks"]).max()))
    np.testing.assert_allclose(mf.cpu().numpy(), g["mask_features"], rtol=0, atol=atol)
    np.testing.assert_allclose(pred["pred_masks"].float().cpu().numpy(), g["pred_masks"], rtol=0, atol=atol * scale)
    np.testing.assert_allclose(pred["reference_points"].cpu().numpy(), g["reference_points"], rtol=0, atol=atol)
    assert len(pred["aux_outputs"]) == int(g["n_aux"])
    for i, a in enumerate(pred["aux_outputs"]):
        np.testing.assert_allclose(a["pred_masks"].float().cpu().numpy(), g["aux%d_pred_masks" % i], rtol=0, atol=atol * scale)


def test_maskformer_head_matches_the_reference_methods(golden, cpu_reference):
    """MaskFormerHead.forward / layers (meta_arch/mask_former_head.py:117-154): pixel decoder -> transformer decoder on its
    three coarse maps and mask features, against the reference's own methods chaining the reference's pixel-decoder
    method and decoder class."""
    g = golden("dec_head_l4")
    head = _head("cpu")
    with torch.no_grad():
        pred, mf = head({k: _t(g["feat_" + k]) for k in _PIX_CH})
    _check_head(g, pred, mf, 2e-4)


@pytest.mark.gpu
def test_maskformer_head_on_the_hip_kernels_matches_the_reference_methods(golden):
    g = golden("dec_head_l4")
    head = _head("cuda")
    with torch.no_grad():
        pred, mf = head({k: _t(g["feat_" + k]).cuda() for k in _PIX_CH})
    _check_head(g, pred, mf, 5e-4)


# ---------------------------------------------------------------- the bf16 autocast path that bench.py times -----------------
def _bf16_outputs_vs_reference(g32, g16, out, what):
    """All-element bound of a bf16-autocast run against the REFERENCE's fp32 outputs, with the reference's own bf16-autocast
    run (same classes, CPU autocast policy; *_bf16.npz) as the yardstick: every prediction head within
    max(1.5 x the reference's own bf16 deviation, 1e-2 x scale), never beyond the absolute cap; the sign of the mask logits
    (what attention masks and instances are cut from) agrees at least as often as the reference's bf16 run does, minus 1 %."""
    scale = max(1.0, float(np.abs(g32["pred_masks"]).max()))
    names = ["pred_masks"] + ["aux%d_pred_masks" % i for i in range(int(g32["n_aux"]))]
    got = [out["pred_masks"]] + [a["pred_masks"] for a in out["aux_outputs"]]
    worst = 0.0
    for name, t in zip(names, got):
        a = t.float().cpu().numpy()
        ref32, ref16 = g32[name], g16[name]
        own = float(np.abs(ref16 - ref32).max()) / scale              # the reference's own bf16 deviation
        err = float(np.abs(a - ref32).max()) / scale
        worst = max(worst, err)
        assert err <= max(1.5 * own, 1e-2), "%s %s: max |err| %.4f x scale, reference bf16 itself %.4f" % (what, name, err, own)
        agree = float(np.mean((a > 0) == (ref32 > 0)))
        own_agree = float(np.mean((ref16 > 0) == (ref32 > 0)))
        assert agree >= own_agree - 0.01, "%s %s: sign agreement %.4f, reference bf16 itself %.4f" % (what, name, agree, own_agree)
    rp = float(np.abs(out["reference_points"].float().cpu().numpy() - g32["reference_points"]).max())
    rp_own = float(np.abs(g16["reference_points"] - g32["reference_points"]).max())
    assert rp <= max(1.5 * rp_own, 5e-3), "%s reference points: %.4f (reference bf16 itself %.4f)" % (what, rp, rp_own)
    return worst


@pytest.mark.gpu
def test_whole_decoder_under_bf16_autocast_stays_within_the_references_own_bf16_deviation(golden):
    """The route bench.py takes (torch.autocast('cuda', bfloat16) around the decoder: MFMA masked attention, MFMA dynamic
    mask head, cached bf16 linears) against dec_full_decoder.npz (reference fp32) and dec_full_decoder_bf16.npz."""
    g32, g16 = golden("dec_full_decoder"), golden("dec_full_decoder_bf16")
    d = _full_decoder("cuda")
    xs = [_t(g32["x%d" % i]).cuda() for i in range(3)]
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        out = d(xs, None, _t(g32["mask_features"]).cuda())
    assert out["pred_masks"].dtype == torch.bfloat16
    worst = _bf16_outputs_vs_reference(g32, g16, out, "decoder")
    assert worst <= 0.05                                             # absolute cap, fraction of the largest logit


@pytest.mark.gpu
def test_head_under_bf16_autocast_stays_within_the_references_own_bf16_deviation(golden):
    """MaskFormerHead as bench.py runs it: fp32 pixel decoder on the HIP kernels (msdeformattn.py:314 forces fp32), bf16
    autocast transformer decoder, against dec_head_l4.npz (reference fp32) and dec_head_l4_bf16.npz."""
    g32, g16 = golden("dec_head_l4"), golden("dec_head_l4_bf16")
    head = _head("cuda")
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        pred, mf = head({k: _t(g32["feat_" + k]).cuda() for k in _PIX_CH})
    assert mf.dtype == torch.float32 and pred["pred_masks"].dtype == torch.bfloat16
    np.testing.assert_allclose(mf.cpu().numpy(), g32["mask_features"], rtol=0, atol=5e-4)      # the fp32 stage is not touched
    worst = _bf16_outputs_vs_reference(g32, g16, pred, "head")
    assert worst <= 0.2
