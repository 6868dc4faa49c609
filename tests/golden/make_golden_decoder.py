#!/usr/bin/env python3
"""Golden vectors for the decoder-side math, from the reference's own code.  RUN IN THE DEV CONTAINER ONLY.

`transformer_decoder/mask2former_transformer_decoder.py` of the reference cannot be imported here: its module header
needs detectron2 / fvcore (absent, un-vendored) and its sibling attention.py does not import on torch >= 2.  The
functions that carry the decoder's arithmetic, however, depend on nothing but torch:

    gen_sineembed_for_position   :21-39      inverse_sigmoid          :41-45      MLP                 :249-261
    dynamic_mask_with_coords     :647-697    mask_heads_forward       :699-719    (methods; `self` is only a parameter bag)
    dice_for                     :917-927    compute_locations        :929-942    parse_dynamic_params :944-979
    SelfAttentionLayer           :47-103     CrossAttentionLayer      :105-193    FFNLayer            :195-235
    select_pos_neg_query         :800-860    select_pos_neg_mask      :862-901    (query-contrast bookkeeping)

and the same holds for attention.py once its broken version test (`A or (B and C) < 9`, attention.py:28) is out of the way:

    MultiheadAttention (class)   attention.py:57-177      multi_head_attention_forward   attention.py:180-387

(`_LinearWithBias` is bound to `torch.nn.modules.linear.NonDynamicallyQuantizableLinear`, the symbol the file's own
else-branch imports on torch >= 1.9.)

Likewise the encoder classes of pixel_decoder/msdeformattn.py (whose header needs detectron2 for the pixel decoder below
them) run on the reference's own MSDeformAttn module and transformer.py helpers, both importable:

    MSDeformAttnTransformerEncoderOnly :23-89   ...EncoderLayer :92-131   ...Encoder :134-162

`MSDeformAttnPixelDecoder.forward_features` (pixel_decoder/msdeformattn.py:314-360) is a method whose body touches only torch
`nn` modules once the encoder uses all four feature levels (num_fpn_levels == 0, :257-258: no detectron2 Conv2d / get_norm
is reached): it is taken from the AST and run on a parameter bag holding exactly what the reference constructor builds for
that case (:213-226 `nn.Sequential(nn.Conv2d, nn.GroupNorm(32, conv_dim))` per level, the reference encoder class, the
reference PositionEmbeddingSine), and `MaskFormerHead.forward / layers` (meta_arch/mask_former_head.py:117-154) on a bag
holding that pixel decoder and the reference transformer decoder.

Finally the whole `MultiScaleMaskedTransformerDecoder` (:267-768) is run in eval mode (targets=None) for a configuration
that never touches its detectron2 names (in_channels == hidden_dim, no forced input projection, no semantic head): the
class is taken from the AST with the `@configurable` decorator of `__init__` (it only adds the from_config calling
convention) and the class's registry decorator not applied, and constructed with explicit keyword arguments.

From connectomics/model/arch/maskformer.py (instance post-processing): comput_mmi :349-354, dice_for :392-401 and
mask_post :403-431.  (mask_nms :357-390 uses `np.int`, gone from numpy >= 1.24, and instance_inference imports imageio,
absent: both are ordinary errors here and stay pinned by literal restatements in the tests.)

From connectomics/model/loss: dice_loss :23-42, sigmoid_ce_loss :50-67, calculate_uncertainty :101-115
(maskformer_criterion.py) and batch_dice_loss :15-30, batch_sigmoid_ce_loss :38-62 (matcher.py).

Round 3 -- the criterion's re-id / reference-point losses and the pixel-embedding loss: `SetCriterion.loss_reid_query`
:318-350, `loss_reid_mask` :352-377, `loss_refpoints` :379-396 and `_get_src_permutation_idx` (loss/maskformer_criterion.py;
methods taken from the class body and run with a bare namespace as `self` -- they touch nothing but torch) and
`discriminative_loss` (loss/loss.py:297-355), fed the items the reference's own select_pos_neg_query / select_pos_neg_mask
build; values AND input gradients are stored (loss_criterion.npz).

This script reads the reference file as text, takes exactly those definitions out of its AST, executes them unmodified
in a namespace holding torch / nn / F / math (no stand-ins for the missing libraries are written), feeds them seeded
inputs and stores inputs + outputs as .npz fixtures next to this file.  Nothing of the reference's source is copied
into the repository; only data is written.

    python tests/golden/make_golden_decoder.py [--ref /root/reference]
"""
import argparse
import ast
import math
import os
import types
from typing import Optional

import numpy as np
import torch
from torch import Tensor, nn
from torch.nn import functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
DEC = "connectomics/model/maskformer_block/transformer_decoder/mask2former_transformer_decoder.py"
ATT = "connectomics/model/maskformer_block/transformer_decoder/attention.py"
WANTED_FUNCS = {"gen_sineembed_for_position", "inverse_sigmoid", "dice_for", "compute_locations", "parse_dynamic_params",
                "_get_activation_fn", "select_pos_neg_query", "select_pos_neg_mask"}
WANTED_CLASSES = {"MLP", "SelfAttentionLayer", "CrossAttentionLayer", "FFNLayer"}
WANTED_METHODS = {"dynamic_mask_with_coords", "mask_heads_forward"}      # of MultiScaleMaskedTransformerDecoder


def load_reference_functions(ref_root, mha_class):
    import copy
    import random
    path = os.path.join(ref_root, DEC)
    tree = ast.parse(open(path).read(), filename=path)
    picked = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in WANTED_FUNCS:
            picked.append(node)
        elif isinstance(node, ast.ClassDef) and node.name in WANTED_CLASSES:
            picked.append(node)
        elif isinstance(node, ast.ClassDef) and node.name == "MultiScaleMaskedTransformerDecoder":
            picked += [n for n in node.body if isinstance(n, ast.FunctionDef) and n.name in WANTED_METHODS]
    # the decoder class itself, `@configurable` (detectron2's from_config calling convention) not applied to __init__
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == "MultiScaleMaskedTransformerDecoder":
            for fn in node.body:
                if isinstance(fn, ast.FunctionDef) and fn.name == "__init__":
                    fn.decorator_list = []
            node.decorator_list = []                             # @TRANSFORMER_DECODER_REGISTRY.register(): fvcore registry
            picked.append(node)
    import importlib.util
    import logging
    spec = importlib.util.spec_from_file_location(
        "ref_position_encoding2", os.path.join(ref_root, os.path.dirname(DEC), "position_encoding.py"))
    pe = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pe)
    ns = {"torch": torch, "nn": nn, "F": F, "math": math, "Tensor": Tensor, "Optional": Optional,
          "MultiheadAttention": mha_class, "random": random, "copy": copy, "logging": logging,
          "PositionEmbeddingSine": pe.PositionEmbeddingSine}
    exec(compile(ast.Module(body=picked, type_ignores=[]), path, "exec"), ns)
    WANTED_CLASSES.add("MultiScaleMaskedTransformerDecoder")
    missing = (WANTED_FUNCS | WANTED_CLASSES | WANTED_METHODS) - set(ns)
    assert not missing, missing
    return types.SimpleNamespace(**{k: ns[k] for k in WANTED_FUNCS | WANTED_CLASSES | WANTED_METHODS})


def load_reference_attention(ref_root):
    import warnings
    from typing import List, Tuple
    from torch.nn.init import constant_, xavier_normal_, xavier_uniform_
    from torch.nn.modules.linear import NonDynamicallyQuantizableLinear
    from torch.overrides import handle_torch_function, has_torch_function
    path = os.path.join(ref_root, ATT)
    tree = ast.parse(open(path).read(), filename=path)
    picked = [n for n in tree.body if (isinstance(n, ast.ClassDef) and n.name == "MultiheadAttention")
              or (isinstance(n, ast.FunctionDef) and n.name == "multi_head_attention_forward")]
    assert len(picked) == 2
    ns = {"torch": torch, "nn": nn, "F": F, "math": math, "Tensor": Tensor, "Optional": Optional, "Tuple": Tuple,
          "List": List, "Module": nn.Module, "warnings": warnings, "constant_": constant_,
          "xavier_uniform_": xavier_uniform_, "xavier_normal_": xavier_normal_,
          "_LinearWithBias": NonDynamicallyQuantizableLinear, "has_torch_function": has_torch_function,
          "handle_torch_function": handle_torch_function, "linear": F.linear, "pad": F.pad, "softmax": F.softmax,
          "dropout": F.dropout}
    exec(compile(ast.Module(body=picked, type_ignores=[]), path, "exec"), ns)
    return ns["MultiheadAttention"]


def load_reference_encoder(ref_root):
    import importlib.util
    import sys
    from torch.nn.init import constant_, normal_, uniform_, xavier_uniform_
    sys.path.insert(0, HERE)
    import make_golden                                            # the MSDeformAttn loader of the op-level generator
    _, mod, _ = make_golden.load_reference(ref_root)
    spec = importlib.util.spec_from_file_location(
        "ref_transformer", os.path.join(ref_root, os.path.dirname(DEC), "transformer.py"))
    tr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tr)
    path = os.path.join(ref_root, "connectomics/model/maskformer_block/pixel_decoder/msdeformattn.py")
    names = {"MSDeformAttnTransformerEncoderOnly", "MSDeformAttnTransformerEncoderLayer", "MSDeformAttnTransformerEncoder"}
    tree = ast.parse(open(path).read(), filename=path)
    picked = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name in names]
    assert len(picked) == 3
    ns = {"torch": torch, "nn": nn, "F": F, "MSDeformAttn": mod.MSDeformAttn, "_get_clones": tr._get_clones,
          "_get_activation_fn": tr._get_activation_fn, "xavier_uniform_": xavier_uniform_, "constant_": constant_,
          "uniform_": uniform_, "normal_": normal_}
    exec(compile(ast.Module(body=picked, type_ignores=[]), path, "exec"), ns)
    return ns["MSDeformAttnTransformerEncoderOnly"]


def load_reference_pixel_decoder_methods(ref_root):
    """forward_features of MSDeformAttnPixelDecoder and forward / layers of MaskFormerHead, as plain functions."""
    from torch.cuda.amp import autocast
    out = {}
    for rel, cls, names in (("connectomics/model/maskformer_block/pixel_decoder/msdeformattn.py", "MSDeformAttnPixelDecoder",
                             {"forward_features"}),
                            ("connectomics/model/maskformer_block/meta_arch/mask_former_head.py", "MaskFormerHead",
                             {"forward", "layers"})):
        path = os.path.join(ref_root, rel)
        tree = ast.parse(open(path).read(), filename=path)
        picked = []
        for node in tree.body:
            if isinstance(node, ast.ClassDef) and node.name == cls:
                picked = [n for n in node.body if isinstance(n, ast.FunctionDef) and n.name in names]
        assert len(picked) == len(names), (cls, names)
        ns = {"torch": torch, "nn": nn, "F": F, "np": np, "autocast": autocast}
        exec(compile(ast.Module(body=picked, type_ignores=[]), path, "exec"), ns)
        out.update({cls + "." + k: ns[k] for k in names})
    return out


def load_reference_postprocessing(ref_root):
    path = os.path.join(ref_root, "connectomics/model/arch/maskformer.py")
    names = {"comput_mmi", "dice_for", "mask_post"}
    tree = ast.parse(open(path).read(), filename=path)
    picked = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(picked) == 3
    ns = {"torch": torch, "np": np}
    exec(compile(ast.Module(body=picked, type_ignores=[]), path, "exec"), ns)
    return types.SimpleNamespace(**{k: ns[k] for k in names})


def load_reference_losses(ref_root):
    out = {}
    for rel, names in (("connectomics/model/loss/maskformer_criterion.py", {"dice_loss", "sigmoid_ce_loss", "calculate_uncertainty"}),
                       ("connectomics/model/loss/matcher.py", {"batch_dice_loss", "batch_sigmoid_ce_loss"})):
        path = os.path.join(ref_root, rel)
        tree = ast.parse(open(path).read(), filename=path)
        picked = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
        assert len(picked) == len(names)
        ns = {"torch": torch, "F": F, "nn": nn}
        exec(compile(ast.Module(body=picked, type_ignores=[]), path, "exec"), ns)
        out.update({k: ns[k] for k in names})
    return types.SimpleNamespace(**out)


def load_reference_criterion(ref_root):
    """The torch-only loss bodies of SetCriterion (methods, `self` is a bare namespace) and discriminative_loss."""
    path = os.path.join(ref_root, "connectomics/model/loss/maskformer_criterion.py")
    names = {"loss_reid_query", "loss_reid_mask", "loss_refpoints", "_get_src_permutation_idx"}
    tree = ast.parse(open(path).read(), filename=path)
    picked = []
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == "SetCriterion":
            picked += [n for n in node.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(picked) == len(names)
    ns = {"torch": torch, "F": F, "nn": nn}
    exec(compile(ast.Module(body=picked, type_ignores=[]), path, "exec"), ns)
    path2 = os.path.join(ref_root, "connectomics/model/loss/loss.py")
    tree2 = ast.parse(open(path2).read(), filename=path2)
    picked2 = [n for n in tree2.body if isinstance(n, ast.FunctionDef) and n.name == "discriminative_loss"]
    assert len(picked2) == 1
    ns2 = {"torch": torch, "F": F, "nn": nn}
    exec(compile(ast.Module(body=picked2, type_ignores=[]), path2, "exec"), ns2)
    bag = types.SimpleNamespace()
    bag._get_src_permutation_idx = types.MethodType(ns["_get_src_permutation_idx"], bag)
    for k in ("loss_reid_query", "loss_reid_mask", "loss_refpoints"):
        setattr(bag, k, types.MethodType(ns[k], bag))
    bag.discriminative_loss = ns2["discriminative_loss"]
    return bag


def save(name, **arrays):
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **{k: np.asarray(v) for k, v in arrays.items()})
    print("wrote", name, {k: np.asarray(v).shape for k, v in arrays.items()})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    RefMHA = load_reference_attention(args.ref)
    ref = load_reference_functions(args.ref, RefMHA)
    g = torch.Generator().manual_seed(2024)

    # ---- projection-free multi-head attention (cross-attention geometry of the PCTrans decoder) ---------------------
    for tag, E, vd, heads, L, S, N in (("ca", 256, 128, 8, 5, 12, 2), ("sa", 128, 128, 8, 6, 6, 3)):
        torch.manual_seed(11)
        mha = RefMHA(E, heads, dropout=0.0, vdim=vd).eval()
        with torch.no_grad():
            mha.out_proj.weight.normal_(0, 0.1)
            mha.out_proj.bias.normal_(0, 0.1)
        q, k, v = torch.randn(L, N, E, generator=g), torch.randn(S, N, E, generator=g), torch.randn(S, N, vd, generator=g)
        bmask = torch.rand(N, 1, L, S, generator=g) < 0.4
        bmask[..., 0] = False                                   # no fully masked row
        bmask = bmask.expand(N, heads, L, S).reshape(N * heads, L, S)
        fmask = torch.randn(L, S, generator=g)
        kpm = torch.zeros(N, S, dtype=torch.bool)
        kpm[:, -2:] = True
        with torch.no_grad():
            o0, w0 = mha(q, k, v)
            o1, w1 = mha(q, k, v, attn_mask=bmask)
            o2, w2 = mha(q, k, v, attn_mask=fmask, key_padding_mask=kpm)
        save("dec_attention_" + tag, q=q, k=k, v=v, heads=heads, out_w=mha.out_proj.weight.detach(),
             out_b=mha.out_proj.bias.detach(), bool_mask=bmask.numpy(), float_mask=fmask, key_padding_mask=kpm.numpy(),
             out_plain=o0, w_plain=w0, out_bool=o1, w_bool=w1, out_float_kpm=o2, w_float_kpm=w2)

    # ---- sine embedding of reference points + inverse sigmoid -------------------------------------------------------
    pts2 = torch.rand(7, 3, 2, generator=g)
    pts4 = torch.rand(5, 2, 4, generator=g)
    xs = torch.cat([torch.tensor([0.0, 1.0, 1e-4, 1e-3, 0.999, 0.9995, -0.2, 1.3]), torch.rand(24, generator=g)])
    save("dec_sineembed_inverse_sigmoid", pts2=pts2, emb2=ref.gen_sineembed_for_position(pts2),
         pts4=pts4, emb4=ref.gen_sineembed_for_position(pts4), emb2_t10=ref.gen_sineembed_for_position(pts2, 10),
         x=xs, inv=ref.inverse_sigmoid(xs), inv_eps1e5=ref.inverse_sigmoid(xs, 1e-5))

    # ---- MLP (the controller / box-embed shape) --------------------------------------------------------------------
    torch.manual_seed(7)
    mlp = ref.MLP(16, 24, 233, 3)
    x = torch.randn(4, 3, 16, generator=g)
    sd = {k: v.detach().numpy() for k, v in mlp.state_dict().items()}
    save("dec_mlp", x=x, y=mlp(x).detach(), **{"sd." + k: v for k, v in sd.items()})

    # ---- dynamic mask head: controller params -> per-query 3-layer conv on [rel coords | mask features] --------------
    for tag, rel, (H, W), tgt in (("rel", True, (12, 10), (6, 5)), ("norel", False, (9, 11), (5, 6)),
                                 ("rel_up", True, (8, 8), (16, 16))):
        N, Q, C, ch, heads, stride = 2, 3, 16, 8, 8, 4
        cin = C + 2 if rel else C
        weight_nums, bias_nums = [cin * ch, ch * ch, ch], [ch, ch, 1]
        bag = types.SimpleNamespace(dynamic_mask_channels=ch, weight_nums=weight_nums, bias_nums=bias_nums,
                                    num_heads=heads)
        bag.mask_heads_forward = types.MethodType(ref.mask_heads_forward, bag)
        feats = torch.randn(N, C, H, W, generator=g)
        refpts = torch.rand(Q, N, 2, generator=g)
        params = torch.randn(Q, N, sum(weight_nums) + sum(bias_nums), generator=g) * 0.3
        logits, amask = ref.dynamic_mask_with_coords(bag, feats, refpts, params, stride, rel, tgt)
        save("dec_dynamic_mask_head_" + tag, feats=feats, refpts=refpts, params=params, stride=stride,
             rel_coord=int(rel), target=np.asarray(tgt), heads=heads, logits_x2=logits, attn_mask=amask.numpy(),
             locations=ref.compute_locations(H, W, stride, "cpu"))

    # ---- the three decoder layer classes (post-norm, eval, dropout 0), position-guided cross-attention ----------------
    d, heads, Q, N, HW = 64, 8, 6, 2, 20
    tgt = torch.randn(Q, N, d, generator=g)
    qpos = torch.randn(Q, N, d, generator=g)
    mem = torch.randn(HW, N, d, generator=g)
    pos = torch.randn(HW, N, d, generator=g)
    qsine = torch.randn(Q, N, 2 * d, generator=g)
    mmask = torch.rand(N, 1, Q, HW, generator=g) < 0.3
    mmask[..., 0] = False
    mmask = mmask.expand(N, heads, Q, HW).reshape(N * heads, Q, HW)
    torch.manual_seed(21)
    sa = ref.SelfAttentionLayer(d, heads).eval()
    ca = ref.CrossAttentionLayer(d, heads).eval()
    ffn = ref.FFNLayer(d, 2 * d).eval()
    with torch.no_grad():
        for m in (sa, ca, ffn):
            for n_, p_ in m.named_parameters():
                if p_.dim() == 1:
                    p_.normal_(0, 0.1) if "norm" not in n_ or "bias" in n_ else p_.uniform_(0.5, 1.5)
        out = {"sa_out": sa(tgt, query_pos=qpos),
               "ca_first": ca(tgt, mem, memory_mask=mmask, pos=pos, query_pos=qpos, query_sine_embed=qsine, is_first=True),
               "ca_later": ca(tgt, mem, memory_mask=mmask, pos=pos, query_pos=qpos, query_sine_embed=qsine, is_first=False),
               "ca_nomask": ca(tgt, mem, pos=pos, query_pos=qpos, query_sine_embed=qsine, is_first=False),
               "ffn_out": ffn(tgt)}
    sds = {}
    for name, m in (("sa", sa), ("ca", ca), ("ffn", ffn)):
        sds.update({"sd.%s.%s" % (name, k): v.detach().numpy() for k, v in m.state_dict().items()})
    save("dec_layers", tgt=tgt, query_pos=qpos, memory=mem, pos=pos, query_sine_embed=qsine, memory_mask=mmask.numpy(),
         heads=heads, **{k: v.detach() for k, v in out.items()}, **sds)

    # ---- query-contrast selection (deterministic parts: 'contrast' and 'label' of every item) -------------------------
    Qc, Nc, C = 14, 2, 8
    query = torch.randn(Qc, Nc, C, generator=g)
    emb = torch.nn.functional.normalize(torch.randn(Nc, Qc, C, generator=g), dim=2)
    emb_dist = emb @ emb.transpose(1, 2)
    masks = torch.randn(Nc, Qc, 6, 5, generator=g) * 2
    pos_indices = [(torch.tensor([1, 5, 9]), torch.tensor([0, 1, 2])), (torch.tensor([0, 13]), torch.tensor([1, 0]))]
    items_q = ref.select_pos_neg_query(query, emb_dist, pos_indices)
    items_m = ref.select_pos_neg_mask(masks, emb_dist, pos_indices)
    arrays = {"query": query, "emb_dist": emb_dist, "masks": masks, "n_items_q": len(items_q), "n_items_m": len(items_m)}
    for b, (src, tg) in enumerate(pos_indices):
        arrays["pos_src_%d" % b], arrays["pos_tgt_%d" % b] = src, tg
    for i, it in enumerate(items_q):
        arrays["q%d_contrast" % i], arrays["q%d_label" % i] = it["contrast"], it["label"]
    for i, it in enumerate(items_m):
        arrays["m%d_contrast" % i], arrays["m%d_label" % i] = it["contrast"], it["label"]
    save("dec_query_contrast", **arrays)

    # ---- MSDeformAttn encoder (level embedding, reference points, 2 layers of deformable self-attention + FFN) --------
    RefEncoder = load_reference_encoder(args.ref)
    torch.manual_seed(31)
    enc = RefEncoder(d_model=32, nhead=4, num_encoder_layers=2, dim_feedforward=64, dropout=0.0, activation="relu",
                     num_feature_levels=3, enc_n_points=4).eval()
    with torch.no_grad():
        for layer in enc.encoder.layers:                          # leave the all-zero offset / attention init
            layer.self_attn.sampling_offsets.weight.normal_(0, 0.05)
            layer.self_attn.attention_weights.weight.normal_(0, 0.3)
            for n_, p_ in layer.named_parameters():
                if p_.dim() == 1 and "norm" in n_:
                    p_.uniform_(0.5, 1.5) if n_.endswith("weight") else p_.normal_(0, 0.1)
    shapes = [(3, 4), (6, 8), (12, 16)]
    srcs = [torch.randn(2, 32, h, w, generator=g) for h, w in shapes]
    poss = [torch.randn(2, 32, h, w, generator=g) for h, w in shapes]
    with torch.no_grad():
        memory, sshapes, starts = enc(srcs, poss)
    arrays = {"memory": memory, "spatial_shapes": sshapes, "level_start_index": starts}
    for i in range(3):
        arrays["src%d" % i], arrays["pos%d" % i] = srcs[i], poss[i]
    arrays.update({"sd." + k: v.detach().numpy() for k, v in enc.state_dict().items()})
    save("dec_msdeform_encoder", **arrays)

    # ---- the whole transformer decoder, eval mode: 3 layers, 6 queries, 3 feature levels + mask features -------------
    # (hidden_dim must be 128: gen_sineembed_for_position hard-codes 128 frequencies.)  The parameters are a
    # deterministic function of their names (tests/golden_params.py), so only inputs and outputs are stored.
    import sys
    sys.path.insert(0, os.path.dirname(HERE))
    from golden_params import deterministic_fill
    decoder = ref.MultiScaleMaskedTransformerDecoder(
        128, True, hidden_dim=128, num_queries=6, nheads=8, dim_feedforward=256, dec_layers=3, pre_norm=False,
        mask_dim=16, enforce_input_project=False, points_num=1, sem_loss_on=False, norm="GN", rel_coord=True).eval()
    deterministic_fill(decoder, 41)
    xs = [torch.randn(2, 128, h, w, generator=g) for h, w in ((2, 3), (4, 5), (8, 10))]
    mfeat = torch.randn(2, 128, 16, 20, generator=g)
    with torch.no_grad():
        out = decoder(xs, None, mfeat)
    arrays = {"x0": xs[0], "x1": xs[1], "x2": xs[2], "mask_features": mfeat, "pred_masks": out["pred_masks"],
              "reference_points": out["reference_points"], "n_aux": len(out["aux_outputs"]),
              "param_names": np.asarray(sorted(decoder.state_dict()))}
    for i, a in enumerate(out["aux_outputs"]):
        arrays["aux%d_pred_masks" % i] = a["pred_masks"]
    for i, a in enumerate(out["aux_reference_points"]):
        arrays["aux%d_reference_points" % i] = a["reference_points"]
    save("dec_full_decoder", **arrays)

    # ---- pixel decoder glue (input projections + GroupNorm, sine PE, encoder, split per level) and the head ------------
    # 4 encoder levels (the north-star geometry) => num_fpn_levels == 0; conv_dim 128 / 8 heads as configured.
    import importlib.util
    meth = load_reference_pixel_decoder_methods(args.ref)
    spec = importlib.util.spec_from_file_location(
        "ref_position_encoding3", os.path.join(args.ref, os.path.dirname(DEC), "position_encoding.py"))
    pe_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pe_mod)
    from golden_params import fill_pixel_decoder
    conv_dim, chans = 128, {"res2": 16, "res3": 24, "res4": 32, "res5": 40}
    feat_hw = {"res2": (16, 20), "res3": (8, 10), "res4": (4, 5), "res5": (2, 3)}

    class PixelDecoderBag(nn.Module):       # the attributes the reference constructor sets for this case (:197-258)
        def __init__(self):
            super().__init__()
            self.in_features = ["res2", "res3", "res4", "res5"]
            self.transformer_in_features = ["res2", "res3", "res4", "res5"]
            self.transformer_num_feature_levels = 4
            self.input_proj = nn.ModuleList(
                nn.Sequential(nn.Conv2d(chans[f], conv_dim, kernel_size=1), nn.GroupNorm(32, conv_dim))
                for f in self.transformer_in_features[::-1])
            self.transformer = RefEncoder(d_model=conv_dim, dropout=0.0, nhead=8, dim_feedforward=1024,
                                          num_encoder_layers=2, num_feature_levels=4)
            self.pe_layer = pe_mod.PositionEmbeddingSine(conv_dim // 2, normalize=True)
            self.maskformer_num_feature_levels = 3
            self.num_fpn_levels = 0
            self.lateral_convs, self.output_convs = [], []
    PixelDecoderBag.forward_features = meth["MSDeformAttnPixelDecoder.forward_features"]
    pix = PixelDecoderBag().eval()
    fill_pixel_decoder(pix, 51)
    g2 = torch.Generator().manual_seed(2025)      # (its own stream: the fixtures below keep the values they always had)
    feats = {f: torch.randn(2, chans[f], *feat_hw[f], generator=g2) for f in chans}
    with torch.no_grad():
        mask_features, enc_feat, multi = pix.forward_features(feats)
    arrays = {"feat_" + f: v for f, v in feats.items()}
    arrays.update(mask_features=mask_features, transformer_encoder_features=enc_feat,
                  param_names=np.asarray(sorted(pix.state_dict())))
    for i, m_ in enumerate(multi):
        arrays["multi_scale_%d" % i] = m_
    save("dec_pixel_decoder_l4", **arrays)

    # ---- the SHIPPED geometry: three encoder levels (res3..res5) + one FPN level for res2 (msdeformattn.py:255-290, 340-350;
    # configs/CVPPP/CVPPP-PCTrans.yaml:17-26).  The loop of forward_features -- lateral conv of res2, bilinear resize of it DOWN
    # to the finest encoder map, sum, output conv, output ordering -- is the reference's own method body (AST, as above).  Its
    # two convolution modules are detectron2 `Conv2d` objects in the reference (`Conv2d(..., bias=use_bias, norm=get_norm(norm,
    # conv_dim)[, activation=F.relu])`, :262-277); detectron2 is not installed, so the wrapper below is WRITTEN HERE from
    # detectron2's documented semantics (an nn.Conv2d that applies `norm`, then `activation`, to its output) with GroupNorm(32)
    # = get_norm("GN"): this fixture pins the reference's loop, resize direction and output order, not detectron2.
    class ConvNormAct(nn.Conv2d):
        def __init__(self, *a, norm=None, activation=None, **k):
            super().__init__(*a, **k)
            self.norm, self.activation = norm, activation

        def forward(self, x):
            x = F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)
            if self.norm is not None:
                x = self.norm(x)
            return self.activation(x) if self.activation is not None else x

    class PixelDecoderBag3(nn.Module):      # the attributes the reference constructor sets for this case (:197-290)
        def __init__(self):
            super().__init__()
            self.in_features = ["res2", "res3", "res4", "res5"]
            self.transformer_in_features = ["res3", "res4", "res5"]
            self.transformer_num_feature_levels = 3
            self.input_proj = nn.ModuleList(
                nn.Sequential(nn.Conv2d(chans[f], conv_dim, kernel_size=1), nn.GroupNorm(32, conv_dim))
                for f in self.transformer_in_features[::-1])
            self.transformer = RefEncoder(d_model=conv_dim, dropout=0.0, nhead=8, dim_feedforward=1024,
                                          num_encoder_layers=2, num_feature_levels=3)
            self.pe_layer = pe_mod.PositionEmbeddingSine(conv_dim // 2, normalize=True)
            self.maskformer_num_feature_levels = 3
            self.num_fpn_levels = 1                       # int(log2(8) - log2(4)), :257-258
            self.adapter_1 = ConvNormAct(chans["res2"], conv_dim, kernel_size=1, bias=False, norm=nn.GroupNorm(32, conv_dim))
            self.layer_1 = ConvNormAct(conv_dim, conv_dim, kernel_size=3, stride=1, padding=1, bias=False,
                                       norm=nn.GroupNorm(32, conv_dim), activation=F.relu)
            self.lateral_convs, self.output_convs = [self.adapter_1], [self.layer_1]
    PixelDecoderBag3.forward_features = meth["MSDeformAttnPixelDecoder.forward_features"]
    pix3 = PixelDecoderBag3().eval()
    fill_pixel_decoder(pix3, 53)
    g3 = torch.Generator().manual_seed(2027)      # (its own stream again)
    feats3 = {f: torch.randn(2, chans[f], *feat_hw[f], generator=g3) for f in chans}
    with torch.no_grad():
        mask_features3, enc_feat3, multi3 = pix3.forward_features(feats3)
    arrays3 = {"feat_" + f: v for f, v in feats3.items()}
    arrays3.update(mask_features=mask_features3, transformer_encoder_features=enc_feat3,
                   param_names=np.asarray(sorted(pix3.state_dict())))
    for i, m_ in enumerate(multi3):
        arrays3["multi_scale_%d" % i] = m_
    save("dec_pixel_decoder_l3_fpn", **arrays3)

    class HeadBag(nn.Module):
        def __init__(self):
            super().__init__()
            self.pixel_decoder = pix
            self.predictor = ref.MultiScaleMaskedTransformerDecoder(
                128, True, hidden_dim=128, num_queries=6, nheads=8, dim_feedforward=256, dec_layers=3, pre_norm=False,
                mask_dim=16, enforce_input_project=False, points_num=1, sem_loss_on=False, norm="GN", rel_coord=True)
            self.transformer_in_feature = "multi_scale_pixel_decoder"
            self.attn_mask_threshold = 0.5
    HeadBag.forward = meth["MaskFormerHead.forward"]
    HeadBag.layers = meth["MaskFormerHead.layers"]
    head = HeadBag().eval()
    deterministic_fill(head.predictor, 41)
    with torch.no_grad():
        pred, mf = head(feats)
    arrays = {"feat_" + f: v for f, v in feats.items()}
    arrays.update(pred_masks=pred["pred_masks"], reference_points=pred["reference_points"], mask_features=mf,
                  n_aux=len(pred["aux_outputs"]))
    for i, a in enumerate(pred["aux_outputs"]):
        arrays["aux%d_pred_masks" % i] = a["pred_masks"]
    save("dec_head_l4", **arrays)

    # ---- instance post-processing helpers (arch/maskformer.py) -------------------------------------------------------
    post = load_reference_postprocessing(args.ref)
    base = (torch.rand(4, 20, 24, generator=g) > 0.6).float()
    inst = torch.cat([base, base[:2] * (torch.rand(2, 20, 24, generator=g) > 0.1).float(),        # near-duplicates
                      (torch.rand(2, 20, 24, generator=g) > 0.7).float()])
    arrays = {"inst_masks": inst, "dice": post.dice_for(inst),
              "post_soft": post.mask_post(inst, thres1=0.5, thres2=0.6, bd_flag=False),
              "post_hard": post.mask_post(inst, thres1=0.5, thres2=0.6, bd_flag=True),
              "post_bbbc": post.mask_post(inst, thres1=0.15, thres2=0.25)}
    mm = [(3.0, 5.0, 2.0), (0.0, 4.0, 0.0), (7.0, 7.0, 7.0)]
    arrays["mmi_in"] = np.asarray(mm, dtype=np.float32)
    arrays["mmi_out"] = np.asarray([float(post.comput_mmi(torch.tensor(a), torch.tensor(b), torch.tensor(c)))
                                    for a, b, c in mm], dtype=np.float32)
    save("arch_mask_post", **arrays)

    # ---- loss / matcher cost definitions ------------------------------------------------------------------------------
    L = load_reference_losses(args.ref)
    logits = torch.randn(5, 300, generator=g) * 2
    tgt = (torch.rand(5, 300, generator=g) > 0.6).float()
    tgt2 = (torch.rand(3, 300, generator=g) > 0.5).float()
    save("loss_functions", logits=logits, targets=tgt, targets2=tgt2, num_masks=3.5,
         dice_loss=L.dice_loss(logits, tgt, 3.5), sigmoid_ce_loss=L.sigmoid_ce_loss(logits, tgt, 3.5),
         uncertainty=L.calculate_uncertainty(logits[:, None, :]), batch_dice=L.batch_dice_loss(logits, tgt2),
         batch_ce=L.batch_sigmoid_ce_loss(logits, tgt2))

    # ---- dice_for (pairwise soft dice of the query masks, used by the query-contrast selection) ---------------------
    m = torch.randn(6, 9, 7, generator=g) * 3
    save("dec_dice_for", masks=m, dice=ref.dice_for(m))

    # ---- round 3: criterion losses on the reference's own query-contrast items (own random stream: every fixture above
    # is reproduced bit for bit) ------------------------------------------------------------------------------------------
    import random
    crit = load_reference_criterion(args.ref)
    g3 = torch.Generator().manual_seed(20261004)
    # 12 queries: a cluster of one positive still takes ALL its negatives into the auxiliary term (10 >= 12 - 2), so the
    # loss does not depend on the iteration order of the reference's Python sets
    Qc, Nc, C = 12, 3, 8
    query = torch.randn(Qc, Nc, C, generator=g3).requires_grad_(True)
    masks = (torch.randn(Nc, Qc, 6, 5, generator=g3) * 2).requires_grad_(True)
    qn = query.detach().permute(1, 0, 2)
    emb_dist = F.cosine_similarity(qn.unsqueeze(2), qn.unsqueeze(1), dim=-1)           # what dec.py:618-620 hands over
    # (every image has a matched query: the reference's argmax over an empty positive list raises)
    pos_indices = [(torch.tensor([1, 5, 9]), torch.tensor([0, 1, 2])), (torch.tensor([0, 11]), torch.tensor([1, 0])),
                   (torch.tensor([4]), torch.tensor([0]))]
    random.seed(5)
    items_q = ref.select_pos_neg_query(query, emb_dist, pos_indices)
    items_m = ref.select_pos_neg_mask(masks, emb_dist, pos_indices)
    lq = crit.loss_reid_query({"pred_qd_query": items_q}, None, None, 1.0)
    lm = crit.loss_reid_mask({"pred_qd_mask": items_m}, None, None, 1.0)
    gq, = torch.autograd.grad(lq["loss_reid_query"] + 0.5 * lq["loss_reid_query_aux"], query)
    gm, = torch.autograd.grad(lm["loss_reid_mask"], masks)
    arrays = {"query": query.detach(), "masks": masks.detach(), "n_items": len(items_q),
              "loss_reid_query": lq["loss_reid_query"].detach(), "loss_reid_query_aux": lq["loss_reid_query_aux"].detach(),
              "loss_reid_mask": lm["loss_reid_mask"].detach(), "grad_query": gq, "grad_masks": gm}
    for b, (src, tg) in enumerate(pos_indices):
        arrays["pos_src_%d" % b], arrays["pos_tgt_%d" % b] = src, tg
    # reference points
    refp = torch.rand(Nc, Qc, 2, generator=g3).requires_grad_(True)
    targets = [{"center_points": torch.rand(n, 1, 2, generator=g3)} for n in (3, 2, 1)]
    lr = crit.loss_refpoints({"reference_points": refp}, targets, pos_indices, 2.5)
    gr, = torch.autograd.grad(lr["loss_refpoints"], refp)
    arrays.update({"reference_points": refp.detach(), "loss_refpoints": lr["loss_refpoints"].detach(), "grad_refpoints": gr,
                   "num_masks": 2.5})
    for b, t in enumerate(targets):
        arrays["center_points_%d" % b] = t["center_points"]
    # pixel-embedding loss: an image without instances, a label that does not occur, instances of one pixel
    emb = torch.randn(3, 8, 12, 10, generator=g3).requires_grad_(True)
    gt = torch.randint(0, 6, (3, 12, 10), generator=g3)
    gt[1] = 0
    gt[2][gt[2] == 3] = 4
    gt[0, 0, 0] = 9
    ld = crit.discriminative_loss(emb, gt)
    ge, = torch.autograd.grad(ld, emb)
    ld2 = crit.discriminative_loss(emb.detach(), gt, delta_v=0.3, delta_d=1.5, alpha=2.0, beta=0.5, gama=0.01)
    arrays.update({"emb": emb.detach(), "seg_gt": gt, "loss_emb": ld.detach(), "grad_emb": ge, "loss_emb_params": ld2})
    save("loss_criterion", **arrays)

    # ---- round 3: the reference decoder / head under bf16 AUTOCAST (what bench.py times): the reference's own classes on
    # the CPU autocast policy (torch.autocast("cpu", bfloat16): linear / conv / matmul / bmm in bf16), same inputs and
    # parameters as dec_full_decoder / dec_head_l4.  The pixel decoder stays fp32 as the reference forces it on the device
    # (msdeformattn.py:314 `@autocast(enabled=False)`): the head's two stages are therefore chained by hand, forward_features
    # outside autocast, predictor inside (mask_former_head.py:141-154).  These are a yardstick, not a bit-level target: they
    # say how far the REFERENCE moves from its own fp32 outputs when its GEMMs run in bf16.
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        out16 = decoder(xs, None, mfeat)
    arrays = {"pred_masks": out16["pred_masks"].float(), "reference_points": out16["reference_points"].float(),
              "n_aux": len(out16["aux_outputs"])}
    for i, a in enumerate(out16["aux_outputs"]):
        arrays["aux%d_pred_masks" % i] = a["pred_masks"].float()
    save("dec_full_decoder_bf16", **arrays)
    with torch.no_grad():
        mf32, _, multi32 = pix.forward_features(feats)
        with torch.autocast("cpu", dtype=torch.bfloat16):
            pred16 = head.predictor(multi32, None, mf32, None, head.attn_mask_threshold, None)
    arrays = {"pred_masks": pred16["pred_masks"].float(), "reference_points": pred16["reference_points"].float(),
              "n_aux": len(pred16["aux_outputs"])}
    for i, a in enumerate(pred16["aux_outputs"]):
        arrays["aux%d_pred_masks" % i] = a["pred_masks"].float()
    save("dec_head_l4_bf16", **arrays)


if __name__ == "__main__":
    main()
