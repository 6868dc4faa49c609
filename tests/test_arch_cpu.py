"""CPU: meta-arch, losses, matcher and instance post-processing ("next" scope).
The reference has no tests for these and cannot import them here, so they are pinned by literal restatements of the
cited lines, optimality / invariance properties, and an end-to-end train step."""
import itertools

import numpy as np
import pytest
import torch

from pctrans_amd.arch import maskformer as mfm
from pctrans_amd.arch.resnet import ResNet
from pctrans_amd.config import get_cfg
from pctrans_amd.loss import Point_HungarianMatcher, SetCriterion
from pctrans_amd.loss import maskformer_criterion as crit
from pctrans_amd.pixel_decoder.ops.modules import ms_deform_attn as msda_mod


@pytest.fixture()
def cpu_reference():
    prev = msda_mod.allow_cpu_reference(True)
    yield
    msda_mod.allow_cpu_reference(prev)


def _blob(h, w, cy, cx, r):
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    return (((ys - cy) ** 2 + (xs - cx) ** 2) <= r * r).float()


def test_matcher_finds_the_planted_assignment_and_is_optimal():
    torch.manual_seed(0)
    H = W = 32
    gts = torch.stack([_blob(H, W, 8, 8, 5), _blob(H, W, 22, 20, 6), _blob(H, W, 10, 24, 4)])
    Q = 6
    pred = torch.randn(Q, H, W) * 0.1 - 4.0
    perm = [4, 0, 2]                                   # query perm[g] predicts gt g
    for g, q in enumerate(perm):
        pred[q] = (gts[g] * 2 - 1) * 6.0
    m = Point_HungarianMatcher(cost_mask=5.0, cost_dice=5.0, num_points=512)
    idx = m({"pred_masks": pred[None]}, [{"masks": gts}])
    src, tgt = idx[0]
    assert sorted(tgt.tolist()) == [0, 1, 2] and len(src) == 3
    assert {int(t): int(s) for s, t in zip(src, tgt)} == {0: 4, 1: 0, 2: 2}
    # optimality against brute force on the same sampled cost matrix (re-seeded so the points coincide)
    torch.manual_seed(1)
    idx = m({"pred_masks": pred[None]}, [{"masks": gts}])
    torch.manual_seed(1)
    pc = torch.rand(1, 512, 2)
    from pctrans_amd.loss.point_features import point_sample
    from pctrans_amd.loss.matcher import batch_dice_loss, batch_sigmoid_ce_loss
    o = point_sample(pred[:, None], pc.repeat(Q, 1, 1), align_corners=False).squeeze(1)
    t = point_sample(gts[:, None], pc.repeat(3, 1, 1), align_corners=False).squeeze(1)
    C = (5.0 * batch_sigmoid_ce_loss(o, t) + 5.0 * batch_dice_loss(o, t)).numpy()
    best = min(sum(C[q, g] for g, q in enumerate(p)) for p in itertools.permutations(range(Q), 3))
    got = sum(C[int(s), int(tt)] for s, tt in zip(*idx[0]))
    assert abs(got - best) < 1e-5


def test_contrast_logsumexp_equals_explicit_double_sum():
    torch.manual_seed(2)
    pred = torch.randn(1, 9)
    label = torch.tensor([[1, 1, 0, 0, 0, 1, 0, 0, 0]])
    got = crit._contrast_logsumexp(pred, label)
    pos, neg = pred[label == 1], pred[label == 0]
    want = torch.log(1 + torch.exp(neg[None, :] - pos[:, None]).sum())
    assert abs(float(got) - float(want)) < 1e-5


def test_discriminative_loss_zero_for_tight_far_clusters_and_positive_otherwise():
    emb = torch.zeros(1, 2, 4, 4)
    gt = torch.zeros(1, 4, 4, dtype=torch.long)
    gt[0, :2] = 1
    gt[0, 2:] = 2
    emb[0, 0, :2] = 10.0
    emb[0, 0, 2:] = -10.0
    # every pixel sits on its centroid: var term = delta_v^2 (the reference's un-hinged form), centroids 20 apart
    l = crit.discriminative_loss(emb, gt)
    assert abs(float(l) - (0.25 + 0.001 * 10.0)) < 1e-5
    assert float(crit.discriminative_loss(torch.randn(1, 2, 4, 4), gt)) > 0


def test_sigmoid_focal_loss_matches_definition():
    x = torch.tensor([[-2.0, 0.0, 3.0]])
    t = torch.tensor([[0.0, 1.0, 1.0]])
    p = torch.sigmoid(x)
    pt = p * t + (1 - p) * (1 - t)
    want = -(0.25 * t + 0.75 * (1 - t)) * (1 - pt) ** 2 * torch.log(pt)
    np.testing.assert_allclose(crit.sigmoid_focal_loss(x, t, alpha=0.25, gamma=2.0).numpy(), want.numpy(), atol=1e-6)


# ---- post-processing -------------------------------------------------------------------------------------------
def _mask_nms_literal(masks, scores, thres):
    """arch/maskformer.py:357-390 restated literally (per-pair sums)."""
    keep, order, nums = [], torch.argsort(scores).tolist()[::-1], masks.shape[0]
    suppressed = np.zeros(nums, dtype=int)
    for i in range(nums):
        idx = order[i]
        if suppressed[idx] == 1:
            continue
        keep.append(idx)
        a = masks[idx]
        for j in range(i, nums):
            jj = order[j]
            if suppressed[jj] == 1:
                continue
            b = masks[jj]
            inter, aa, ab = (a * b).sum(), a.sum(), b.sum()
            if aa == 0 or ab == 0:
                aa, ab = aa + 1e-5, ab + 1e-5
            if max(inter / aa, inter / ab) >= thres:
                suppressed[jj] = 1
    return masks[keep]


def test_mask_nms_and_mask_post_equal_literal_restatement():
    torch.manual_seed(3)
    H = W = 24
    masks = torch.stack([_blob(H, W, 8, 8, 5), _blob(H, W, 9, 8, 5), _blob(H, W, 16, 16, 4), _blob(H, W, 8, 9, 6),
                         _blob(H, W, 17, 16, 4), _blob(H, W, 3, 20, 2)])
    scores = masks.flatten(1).sum(1) / masks.flatten(1).sum(1).max()
    got = mfm.mask_nms(masks, scores, thres=0.72)
    want = _mask_nms_literal(masks, scores, 0.72)
    assert got.shape == want.shape and torch.equal(got, want)
    merged = mfm.mask_post(masks, thres1=0.5, thres2=0.6, bd_flag=True)
    # literal: first-come clustering on dice > thres1, mean, threshold
    d = mfm.dice_for(masks)
    taken, clusters = [], []
    for i in range(6):
        if i in taken:
            continue
        c = torch.where(d[i] > 0.5)[0].tolist()
        taken += c
        clusters.append(c)
    lit = torch.stack([(masks[c].mean(0) > 0.6).float() for c in clusters])
    assert torch.equal(merged, lit) and merged.shape[0] < 6


def test_instance_inference_labels_disjoint_blobs():
    net = mfm.MaskFormer(backbone=torch.nn.Identity(), sem_seg_head=torch.nn.Identity(),
                         criterion=torch.nn.Identity(), num_queries=4, dataset_name="BBBC")
    H = W = 40
    logits = torch.full((5, H, W), -8.0)
    logits[0] = (_blob(H, W, 10, 10, 6) * 2 - 1) * 8
    logits[1] = (_blob(H, W, 10, 10, 6) * 2 - 1) * 8          # duplicate of 0 -> merged by mask_post
    logits[2] = (_blob(H, W, 28, 28, 7) * 2 - 1) * 8
    logits[3] = (_blob(H, W, 30, 8, 2) * 2 - 1) * 8           # area < 40 -> dropped
    out, bd = net.instance_inference(logits)
    assert bd is None and out.shape == (1, H, W) and out.dtype == torch.int16
    ids = set(out.unique().tolist())
    assert ids == {0, 1, 2}
    assert int(out[0, 10, 10]) != int(out[0, 28, 28]) and int(out[0, 30, 8]) == 0


# ---- end to end ---------------------------------------------------------------------------------------------------
def _model(Q=6):
    torch.manual_seed(0)
    cfg = get_cfg(num_queries=Q, norm="BN", sem_norm="BN", enc_layers=1, dec_layers=3, train_num_points=256,
                  dataset="BBBC")
    backbone = ResNet(18, in_channels=3, norm="BN")
    return mfm.MaskFormer(**mfm.MaskFormer.from_config(cfg, backbone))


def test_maskformer_train_step_and_eval(cpu_reference):
    import random
    random.seed(0)
    model = _model()
    assert {"backbone", "sem_seg_head", "criterion"} <= {n for n, _ in model.named_children()}
    H = W = 64
    vol = torch.randn(2, 3, H, W)
    targets = []
    for b in range(2):
        masks = torch.stack([_blob(H, W, 16, 16, 8), _blob(H, W, 44, 40, 10)])
        fg = (masks.sum(0) > 0).float()
        centers = torch.tensor([[16 / W, 16 / H], [40 / W, 44 / H]]).view(2, 1, 2)
        targets.append({"masks": masks, "labels": torch.ones(2, dtype=torch.long), "fg_masks": fg,
                        "center_points": centers})
    model.train()
    losses = model(vol, targets, True)
    assert "loss_mask" in losses and "loss_dice_1" in losses and "loss_refpoints" in losses and "loss_sem" in losses
    total = sum(v for v in losses.values() if torch.is_tensor(v))
    assert torch.isfinite(total)
    total.backward()
    g = model.sem_seg_head.predictor.controller.layers[0].weight.grad
    assert g is not None and torch.isfinite(g).all() and float(g.abs().sum()) > 0
    assert model.sem_seg_head.pixel_decoder.transformer.encoder.layers[0].self_attn.value_proj.weight.grad is not None
    model.eval()
    with torch.no_grad():
        out, bd = model(vol)
    assert out.shape == (2, H, W) and out.dtype == torch.int16 and bd is None


# ---- batched forms of the reference's per-item loops (one pass per batch instead of a dozen launches per item) ----------
def _discriminative_loss_loop(embedding, seg_gt, delta_v=0.5, delta_d=3, alpha=1, beta=1, gama=0.001):
    """Literal restatement of loss/loss.py:297-355 of the reference: one pass per instance label."""
    import torch.nn.functional as F
    bs, ed = embedding.shape[0], embedding.shape[1]
    var_loss = dist_loss = reg_loss = embedding.new_zeros(())
    for b in range(bs):
        emb_b, gt_b = embedding[b], seg_gt[b]
        labels = torch.unique(gt_b)
        labels = labels[labels != 0]
        num_id = len(labels)
        if num_id == 0:
            continue
        centroids = []
        for idx in labels:
            emb_i = emb_b[:, gt_b == idx]
            mean_i = emb_i.mean(dim=1)
            centroids.append(mean_i)
            var_loss = var_loss + torch.mean((torch.norm(emb_i - mean_i.reshape(ed, 1), dim=0) - delta_v) ** 2) / num_id
        centroids = torch.stack(centroids)
        if num_id > 1:
            d = torch.norm(centroids.reshape(-1, 1, ed) - centroids.reshape(1, -1, ed), dim=2)
            d = d + torch.eye(num_id) * delta_d
            dist_loss = dist_loss + torch.sum(F.relu(-d + delta_d) ** 2) / (num_id * (num_id - 1)) / 2
        reg_loss = reg_loss + torch.mean(torch.norm(centroids, dim=1))
    return alpha * var_loss / bs + beta * dist_loss / bs + gama * reg_loss / bs


def test_discriminative_loss_equals_the_per_instance_loop():
    torch.manual_seed(3)
    emb = torch.randn(3, 8, 12, 10, requires_grad=True)
    gt = torch.randint(0, 6, (3, 12, 10))
    gt[1] = 0                                              # an image without instances
    gt[2][gt[2] == 3] = 4                                  # a label that is skipped
    got = crit.discriminative_loss(emb, gt)
    want = _discriminative_loss_loop(emb, gt)
    assert abs(float(got.detach()) - float(want.detach())) < 1e-5 * max(1.0, abs(float(want.detach())))
    g1, = torch.autograd.grad(got, emb)
    g2, = torch.autograd.grad(want, emb)
    torch.testing.assert_close(g1, g2, rtol=1e-4, atol=1e-6)


def test_reid_losses_batched_equal_the_per_item_loops():
    """SetCriterion.loss_reid_query / loss_reid_mask on ContrastItems (all items in one pass) against the same items as
    a plain list (the reference's loop, maskformer_criterion.py:300-365), values and gradients."""
    import random
    from pctrans_amd.transformer_decoder import query_contrast as qc
    torch.manual_seed(4)
    Q, N, C = 14, 3, 16
    output = torch.randn(Q, N, C, requires_grad=True)
    masks = torch.randn(N, Q, 6, 5, requires_grad=True)
    indices = [(torch.tensor([1, 4, 7]), torch.tensor([0, 1, 2])), (torch.tensor([0, 13]), torch.tensor([1, 0])),
               (torch.tensor([], dtype=torch.long), torch.tensor([], dtype=torch.long))]
    random.seed(7)
    items_q, items_m = qc.query_contrast_items(output, masks, indices)
    assert items_q.batched is not None and len(items_q) == len(items_m) > 2
    c = SetCriterion(1, None, {}, 0.1, [], 16, 3, 0.75)
    fast = {**c.loss_reid_query({"pred_qd_query": items_q}, None, None, 1),
            **c.loss_reid_mask({"pred_qd_mask": items_m}, None, None, 1)}
    slow = {**c.loss_reid_query({"pred_qd_query": list(items_q)}, None, None, 1),
            **c.loss_reid_mask({"pred_qd_mask": list(items_m)}, None, None, 1)}
    assert set(fast) == {"loss_reid_query", "loss_reid_query_aux", "loss_reid_mask"}
    for k in fast:
        assert abs(float(fast[k].detach()) - float(slow[k].detach())) < 1e-5 * max(1.0, abs(float(slow[k].detach()))), k
    gf = torch.autograd.grad(sum(fast.values()), [output, masks], retain_graph=True)
    gs = torch.autograd.grad(sum(slow.values()), [output, masks])
    for a, b in zip(gf, gs):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-6)
    # an item whose scores overflow exp(): the factorised form stays finite like the reference's logsumexp
    big = torch.tensor([[300.0, -300.0, 250.0]])
    assert abs(float(crit._contrast_softplus(big, torch.tensor([1]))) -
               float(crit._contrast_logsumexp(big, torch.tensor([[1, 0, 0]])))) < 1e-3


def test_matcher_batched_costs_equal_the_per_image_loop():
    torch.manual_seed(5)
    H = W = 24
    Q = 7
    pred = torch.randn(3, Q, H // 2, W // 2) * 3
    targets = [{"masks": (torch.rand(g, H, W) < 0.3).float()} for g in (3, 5, 1)]
    m = Point_HungarianMatcher(cost_mask=5.0, cost_dice=2.0, num_points=64)
    torch.manual_seed(9)
    fast = m({"pred_masks": pred}, targets)
    m.batch_images = False
    torch.manual_seed(9)
    slow = m({"pred_masks": pred}, targets)
    for (a, b), (c, d) in zip(fast, slow):
        assert a.tolist() == c.tolist() and b.tolist() == d.tolist()
    m.batch_images = True
    torch.manual_seed(9)
    C, counts = m._batched_costs({"pred_masks": pred}, targets)
    assert counts == [3, 5, 1] and C.shape == (3, Q, 5)
    # ragged target sizes: falls back to the loop
    targets[1]["masks"] = (torch.rand(2, H + 2, W) < 0.3).float()
    assert m._batched_costs({"pred_masks": pred}, targets) is None
    assert len(m({"pred_masks": pred}, targets)) == 3


def test_matcher_forward_many_equals_one_call_per_prediction():
    """forward_many (the decoder's ten heads in one call) = forward per prediction with the same random stream; on CPU tensors
    it takes the per-call path."""
    torch.manual_seed(6)
    preds = [torch.randn(2, 9, 12, 12) * 3 for _ in range(4)]
    targets = [{"masks": (torch.rand(g, 24, 24) < 0.3).float()} for g in (3, 5)]
    m = Point_HungarianMatcher(cost_mask=5.0, cost_dice=2.0, num_points=64)
    torch.manual_seed(2)
    one_by_one = [m({"pred_masks": p}, targets) for p in preds]
    torch.manual_seed(2)
    many = m.forward_many([{"pred_masks": p} for p in preds], targets)
    assert len(many) == len(preds)
    for a, b in zip(many, one_by_one):
        assert len(a) == len(b) == 2
        for (i, j), (k, l) in zip(a, b):
            assert i.tolist() == k.tolist() and j.tolist() == l.tolist()
    assert m.forward_many([], targets) == []


# ---- instance_inference end to end (arch/maskformer.py:267-346 of the reference) ----------------------------------------
def _instance_inference_literal(mask_pred, dataset):
    """The reference's instance_inference restated line by line for both dataset branches (its CVPPP branch cannot run as
    shipped: it imports imageio, writes prd_result.png and stops in pdb, :284-305), on the reference-pinned mask_post
    (fixture arch_mask_post.npz) and the literal mask_nms above."""
    mask_pred = mask_pred.sigmoid().float()
    if dataset == "CVPPP":
        pred_masks = (mask_pred > 0.69).float()
        areas = torch.tensor([pred_masks[n].sum() for n in range(pred_masks.shape[0])])
        pred_masks = pred_masks[areas > 40]
        pred_masks = mfm.mask_post(pred_masks, thres1=0.5, thres2=0.6, bd_flag=True)
        areas = torch.tensor([pred_masks[n].sum() for n in range(pred_masks.shape[0])])
        pred_masks = _mask_nms_literal(pred_masks, (areas / areas.max()).to(pred_masks), 0.72)
    else:
        pred_masks = (mask_pred > 0.05).float()
        areas = torch.tensor([pred_masks[n].sum() for n in range(pred_masks.shape[0])])
        pred_masks = pred_masks[areas > 40]
        pred_masks = mfm.mask_post(pred_masks, thres1=0.15, thres2=0.25)
    areas = torch.tensor([pred_masks[n].sum() for n in range(pred_masks.shape[0])])
    pred_masks = pred_masks[torch.argsort(areas).tolist()]
    mask_scores = torch.cat([torch.zeros((1, pred_masks.shape[-2], pred_masks.shape[-1])).to(pred_masks), pred_masks])
    return torch.argmax(mask_scores, axis=0).to(torch.int16)[None, :]


def instance_logits(seed=0, H=96, W=88):
    """Query mask logits with what the post-processing has to sort out: near-duplicate queries (merged), nested and
    overlapping instances (NMS on CVPPP), soft edges around the thresholds, specks below the 40-pixel floor, empty
    queries.  Every instance has its own area, so no argsort tie decides a label."""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    specs = [(20, 20, 9.0), (20.5, 20.2, 9.3), (60, 30, 12.0), (60, 34, 6.5), (30, 60, 7.5), (75, 70, 10.5), (76, 71, 10.0),
             (45, 45, 5.0), (10, 75, 3.0), (88, 10, 2.0), (50, 80, 8.2)]
    logits = torch.full((len(specs) + 3, H, W), -9.0)
    for i, (cy, cx, r) in enumerate(specs):
        d = torch.sqrt((yy - cy) ** 2 + (xx - cx) ** 2)
        logits[i] = (r - d) * 1.7 + 0.05 * torch.randn(H, W, generator=g)      # soft edge: sigmoid crosses 0.05 .. 0.69 over ~3 px
    logits[-1] = -3.0 + 0.3 * torch.randn(H, W, generator=g)                     # a faint query: above 0.05 only in places
    return logits


@pytest.mark.parametrize("dataset", ["CVPPP", "BBBC"])
def test_instance_inference_equals_the_literal_restatement(dataset):
    net = mfm.MaskFormer(backbone=torch.nn.Identity(), sem_seg_head=torch.nn.Identity(),
                         criterion=torch.nn.Identity(), num_queries=4, dataset_name=dataset)
    logits = instance_logits()
    out, bd = net.instance_inference(logits)
    want = _instance_inference_literal(logits, dataset)
    assert bd is None and out.dtype == torch.int16 and out.shape == want.shape
    assert torch.equal(out, want)
    assert len(out.unique()) >= 6                      # several instances survive, several queries were merged / dropped
    assert len(out.unique()) - 1 < logits.shape[0] - 3
