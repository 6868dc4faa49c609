"""PCTrans set criterion: point-sampled mask BCE + dice, reference-point L1, query / mask contrast (re-id) losses,
semantic focal loss and discriminative pixel-embedding loss, with deep supervision over the decoder layers.

Mirrors loss/maskformer_criterion.py:118-506 of the reference (`SetCriterion` and its loss_* methods; the helper
functions dice_loss / sigmoid_ce_loss / calculate_uncertainty at :23-116) and loss/loss.py:297-355
(`discriminative_loss`).  `sigmoid_focal_loss` is fvcore's published formula (the reference imports
fvcore.nn.sigmoid_focal_loss_jit, :21).  The world-size normaliser of `num_masks` uses torch.distributed directly
(all_reduce over RCCL / gloo) instead of detectron2.utils.comm.
"""
import torch
import torch.distributed as dist
import torch.nn.functional as F
from torch import nn

from .point_features import get_uncertain_point_coords_with_randomness, point_sample


def dice_loss(inputs, targets, num_masks: float):
    inputs = inputs.sigmoid().flatten(1)
    numerator = 2 * (inputs * targets).sum(-1)
    denominator = inputs.sum(-1) + targets.sum(-1)
    return (1 - (numerator + 1) / (denominator + 1)).sum() / num_masks


def sigmoid_ce_loss(inputs, targets, num_masks: float):
    return F.binary_cross_entropy_with_logits(inputs, targets, reduction="none").mean(1).sum() / num_masks


def calculate_uncertainty(logits):
    assert logits.shape[1] == 1
    return -(torch.abs(logits))


def sigmoid_focal_loss(inputs, targets, alpha: float = -1, gamma: float = 2, reduction: str = "none"):
    p = torch.sigmoid(inputs)
    ce_loss = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    p_t = p * targets + (1 - p) * (1 - targets)
    loss = ce_loss * ((1 - p_t) ** gamma)
    if alpha >= 0:
        loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
    if reduction == "mean":
        loss = loss.mean()
    elif reduction == "sum":
        loss = loss.sum()
    return loss


DISCRIMINATIVE_DENSE_BUDGET = 64 * 1024 * 1024      # pixels x labels above which discriminative_loss loops over instances


def discriminative_loss(embedding, seg_gt, delta_v=0.5, delta_d=3, alpha=1, beta=1, gama=0.001):
    """Pull pixels to their instance centroid, push centroids apart (loss/loss.py:297-355)."""
    batch_size, embed_dim = embedding.shape[0], embedding.shape[1]
    var_loss = embedding.new_zeros(())
    dist_loss = embedding.new_zeros(())
    reg_loss = embedding.new_zeros(())
    for b in range(batch_size):
        emb_b, gt_b = embedding[b].reshape(embed_dim, -1), seg_gt[b].reshape(-1)
        labels, inv = torch.unique(gt_b, return_inverse=True)                   # ascending: the reference's loop order
        keep = labels != 0
        num_id = int(keep.sum())
        if num_id == 0:
            zero = embedding.sum() * 0
            var_loss, dist_loss, reg_loss = var_loss + zero, dist_loss + zero, reg_loss + zero
            continue
        if emb_b.shape[1] * labels.numel() > DISCRIMINATIVE_DENSE_BUDGET:
            # a crowded, large sample (1024^2 pixels x 1000 instances would need ~4 GB for the membership matrix): the
            # reference's per-instance loop, O(pixels) memory per instance
            cents = []
            for idx in labels[keep]:
                emb_i = emb_b[:, gt_b == idx]
                mean_i = emb_i.mean(dim=1)
                cents.append(mean_i)
                var_loss = var_loss + torch.mean((torch.norm(emb_i - mean_i.reshape(embed_dim, 1), dim=0) - delta_v) ** 2) / num_id
            centroids = torch.stack(cents)
        else:
            # all instances at once: membership as a one-hot [pixels, labels] matrix, so centroids and per-instance means
            # are two small matrix products (deterministic, unlike scatter-adds) instead of ~8 launches per instance
            member = F.one_hot(inv, labels.numel()).to(emb_b.dtype)
            count = member.sum(0)
            mean = (emb_b @ member) / count                                      # [E, labels]
            spread = (torch.norm(emb_b - mean[:, inv], dim=0) - delta_v) ** 2    # [pixels]
            var_loss = var_loss + ((spread @ member) / count)[keep].sum() / num_id
            centroids = mean[:, keep].t()                                        # [num_id, E]
        if num_id > 1:
            d = torch.norm(centroids.reshape(-1, 1, embed_dim) - centroids.reshape(1, -1, embed_dim), dim=2)
            d = d + torch.eye(num_id, dtype=d.dtype, device=d.device) * delta_d
            dist_loss = dist_loss + torch.sum(F.relu(-d + delta_d) ** 2) / (num_id * (num_id - 1)) / 2
        reg_loss = reg_loss + torch.mean(torch.norm(centroids, dim=1))
    return alpha * var_loss / batch_size + beta * dist_loss / batch_size + gama * reg_loss / batch_size


def _pad_masks(masks):
    """Zero-pad a list of [G_i, H_i, W_i] masks to a common [G_i, Hmax, Wmax] (nested_tensor_from_tensor_list)."""
    H = max(m.shape[1] for m in masks)
    W = max(m.shape[2] for m in masks)
    C = max(m.shape[0] for m in masks)
    out = masks[0].new_zeros((len(masks), C, H, W))
    for i, m in enumerate(masks):
        out[i, :m.shape[0], :m.shape[1], :m.shape[2]].copy_(m)
    return out


def _contrast_logsumexp(pred, label):
    """log(1 + sum_{pos p, neg n} exp(s_n - s_p)) as the reference builds it (:313-327, :347-361)."""
    pos_inds, neg_inds = label == 1, label == 0
    pred_pos = pred * pos_inds.float()
    pred_neg = pred * neg_inds.float()
    pred_pos = pred_pos.masked_fill(neg_inds, float("inf"))
    pred_neg = pred_neg.masked_fill(pos_inds, float("-inf"))
    n = pred.shape[1]
    x = F.pad(pred_neg.repeat(1, n) - torch.repeat_interleave(pred_pos, n, dim=1), (0, 1), "constant", 0)
    return torch.logsumexp(x, dim=1)


def _contrast_softplus(scores, n_pos):
    """The same loss for a batch of items whose first n_pos[i] columns are the positives: the double sum factorises,
    log(1 + sum_n e^{s_n} * sum_p e^{-s_p}) = softplus(logsumexp_neg(s) + logsumexp_pos(-s)).  scores [I, K] -> [I].
    (An item without negatives gives logsumexp over nothing = -inf and a loss of 0, like the reference's lone padded 0.)"""
    pos = torch.arange(scores.shape[1], device=scores.device)[None, :] < n_pos[:, None]
    lse_neg = torch.logsumexp(scores.masked_fill(pos, float("-inf")), dim=1)
    lse_pos = torch.logsumexp((-scores).masked_fill(~pos, float("-inf")), dim=1)
    return F.softplus(lse_neg + lse_pos)


class SetCriterion(nn.Module):
    def __init__(self, num_classes, matcher, weight_dict, eos_coef, losses, num_points, oversample_ratio,
                 importance_sample_ratio):
        super().__init__()
        self.num_classes = num_classes
        self.matcher = matcher
        self.weight_dict = weight_dict
        self.eos_coef = eos_coef
        self.losses = losses
        empty_weight = torch.ones(self.num_classes + 1)
        empty_weight[-1] = self.eos_coef
        self.register_buffer("empty_weight", empty_weight)
        self.num_points = num_points
        self.oversample_ratio = oversample_ratio
        self.importance_sample_ratio = importance_sample_ratio

    def loss_masks(self, outputs, targets, indices, num_masks):
        assert "pred_masks" in outputs
        src_idx = self._get_src_permutation_idx(indices)
        tgt_idx = self._get_tgt_permutation_idx(indices)
        src_masks = outputs["pred_masks"][src_idx]
        target_masks = _pad_masks([t["masks"] for t in targets]).to(src_masks)[tgt_idx]
        src_masks, target_masks = src_masks[:, None], target_masks[:, None]
        with torch.no_grad():
            point_coords = get_uncertain_point_coords_with_randomness(
                src_masks, calculate_uncertainty, self.num_points, self.oversample_ratio, self.importance_sample_ratio)
            point_labels = point_sample(target_masks, point_coords, align_corners=False).squeeze(1)
        point_logits = point_sample(src_masks, point_coords, align_corners=False).squeeze(1)
        return {"loss_mask": sigmoid_ce_loss(point_logits, point_labels, num_masks),
                "loss_dice": dice_loss(point_logits, point_labels, num_masks)}

    def loss_embedding(self, emb, targets, alpha=1, beta=1, gama=0.001):
        gt = []
        for target in targets:
            down = F.interpolate(target["masks"][:, None].to(torch.float), size=[emb.shape[2], emb.shape[3]],
                                 mode="nearest")
            down = torch.cat([down.new_zeros((1, 1, down.shape[-2], down.shape[-1])), down])
            gt.append(torch.argmax((down[:, 0] > 0).to(torch.int16), dim=0))
        return {"loss_emb": discriminative_loss(emb, torch.stack(gt), alpha=alpha, beta=beta, gama=gama)}

    def loss_reid_query(self, outputs, targets, indices, num_masks):
        items = outputs["pred_qd_query"]
        if len(items) == 0:
            return {"loss_reid_query": 0, "loss_reid_query_aux": 0}
        bt = getattr(items, "batched", None)
        if bt is not None:                       # all items at once (query_contrast.ContrastItems)
            contras = _contrast_softplus(bt["contrast"] / 2.0, bt["n_pos"]).sum()
            sq = torch.abs(bt["aux_consin"] - bt["aux_label"]) ** 2
            aux = ((sq[bt["aux_rows"]] * bt["aux_valid"]).sum(1) / bt["aux_count"]).sum()
            return {"loss_reid_query": contras / len(items), "loss_reid_query_aux": aux / len(items)}
        contras, aux = 0, 0
        for it in items:
            contras = contras + _contrast_logsumexp(it["contrast"].permute(1, 0) / 2.0, it["label"].unsqueeze(0))
            aux = aux + (torch.abs(it["aux_consin"].permute(1, 0) - it["aux_label"].unsqueeze(0)) ** 2).mean()
        return {"loss_reid_query": contras.sum() / len(items), "loss_reid_query_aux": aux / len(items)}

    def loss_reid_mask(self, outputs, targets, indices, num_masks):
        items = outputs["pred_qd_mask"]
        if len(items) == 0:
            return {"loss_reid_mask": 0}
        bt = getattr(items, "batched", None)
        if bt is not None:
            return {"loss_reid_mask": _contrast_softplus(bt["contrast"] / 0.5, bt["n_pos"]).sum() / len(items)}
        contras = 0
        for it in items:
            contras = contras + _contrast_logsumexp(it["contrast"].permute(1, 0) / 0.5, it["label"].unsqueeze(0))
        return {"loss_reid_mask": contras.sum() / len(items)}

    def loss_refpoints(self, outputs, targets, indices, num_masks):
        assert "reference_points" in outputs
        idx = self._get_src_permutation_idx(indices)
        src_points = outputs["reference_points"][idx]
        target_points = torch.cat([t["center_points"][i] for t, (_, i) in zip(targets, indices)], dim=0).flatten(1)
        return {"loss_refpoints": F.l1_loss(src_points, target_points, reduction="none").sum() / num_masks}

    def loss_sem(self, outputs, targets):
        assert "sem_mask" in outputs
        logits_pred = outputs["sem_mask"]
        sem = torch.stack([t["fg_masks"] for t in targets])
        out_stride = 8
        one_hot = sem[:, None, out_stride // 2::out_stride, out_stride // 2::out_stride].float()
        num_pos = (one_hot > 0).sum().float().clamp(min=1.0)
        return {"loss_sem": sigmoid_focal_loss(logits_pred, one_hot, alpha=0.25, gamma=2.0, reduction="sum") / num_pos}

    def _get_src_permutation_idx(self, indices):
        batch_idx = torch.cat([torch.full_like(src, i) for i, (src, _) in enumerate(indices)])
        return batch_idx, torch.cat([src for (src, _) in indices])

    def _get_tgt_permutation_idx(self, indices):
        batch_idx = torch.cat([torch.full_like(tgt, i) for i, (_, tgt) in enumerate(indices)])
        return batch_idx, torch.cat([tgt for (_, tgt) in indices])

    def get_loss(self, loss, outputs, targets, indices, num_masks):
        loss_map = {"masks": self.loss_masks, "reid_query": self.loss_reid_query, "reid_mask": self.loss_reid_mask}
        assert loss in loss_map, f"do you really want to compute {loss} loss?"
        return loss_map[loss](outputs, targets, indices, num_masks)

    def forward(self, outputs, targets, pixel_embedding=None):
        indices = outputs["indices_list"][-1]
        num_masks = sum(len(t["labels"]) for t in targets)
        num_masks = torch.as_tensor([num_masks], dtype=torch.float, device=next(iter(outputs.values())).device)
        world = 1
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(num_masks)
            world = dist.get_world_size()
        if hasattr(self.matcher, "check"):
            self.matcher.check()               # device-side assignments: infeasible cost matrices surface here
        num_masks = torch.clamp(num_masks / world, min=1).item()

        losses = {}
        for loss in self.losses:
            if loss == "embedding":
                losses.update(self.loss_embedding(pixel_embedding, targets))
            elif loss == "refpoints":
                losses.update(self.loss_refpoints(outputs, targets, indices, num_masks))
            elif loss == "sem":
                losses.update(self.loss_sem(outputs, targets))
            else:
                losses.update(self.get_loss(loss, outputs, targets, indices, num_masks))

        if "aux_outputs" in outputs:
            for i, aux_outputs in enumerate(outputs["aux_outputs"]):
                indices = outputs["indices_list"][i]
                for loss in self.losses:
                    if "reid" in loss or loss == "sem":
                        continue
                    if not (loss == "embedding" or loss == "refpoints"):
                        l_dict = self.get_loss(loss, aux_outputs, targets, indices, num_masks)
                        losses.update({k + f"_{i}": v for k, v in l_dict.items()})
                    if i != 0:
                        l_dict = self.loss_refpoints(outputs["aux_reference_points"][i - 1], targets, indices,
                                                     num_masks)
                        losses.update({k + f"_{i}": v for k, v in l_dict.items()})
        return losses
