"""MaskFormer meta-architecture of PCTrans: backbone -> MaskFormerHead -> losses (train) or instance maps (eval).

API mirror of connectomics/model/arch/maskformer.py of the reference:
    MaskFormer.forward(volume, targets=None, train=False)            :165-257  (called by engine/trainer.py:151,417,524)
    MaskFormer.from_config(cfg)                                      :72-159   (criterion / weight_dict wiring)
    instance_inference (CVPPP / BBBC post-processing)                :267-346
    comput_mmi, mask_nms, dice_for, mask_post                        :349-431
Sub-module names (`backbone`, `sem_seg_head`, `criterion`) are kept, so checkpoints interchange.

Same results, different execution for the post-processing: per-mask `.sum()` host round trips and
O(Q^2) Python loops over mask pairs are replaced by one [Q, HW] x [HW, Q] GEMM (pairwise intersections / dice) and a
greedy pass over that small matrix; the reference's debugging leftovers in the CVPPP branch (`io.imsave`,
`pdb.set_trace()`, :305-306) are not reproduced.
"""
from typing import List

import torch
from torch import nn
from torch.nn import functional as F

from ..loss import Point_HungarianMatcher, SetCriterion
from ..meta_arch.mask_former_head import MaskFormerHead


def dice_for(inputs):
    """Pairwise dice of (binary) masks [Q, H, W] -> [Q, Q] (:392-401)."""
    x = inputs.flatten(1)
    numerator = x @ x.transpose(-2, -1)
    s = x.sum(-1)
    return (2 * numerator + 1) / (s[:, None] + s[None, :] + 1)


def comput_mmi(area_a, area_b, intersect):
    eps = 0.00001
    if area_a == 0 or area_b == 0:
        area_a, area_b = area_a + eps, area_b + eps
    return max(intersect / area_a, intersect / area_b)


def mask_nms(masks, scores, thres=0.3):
    """Greedy mask NMS on max(intersection / area_a, intersection / area_b) (:357-390)."""
    nums = masks.shape[0]
    if nums == 0:
        return masks
    flat = masks.flatten(1).float()
    inter = (flat @ flat.t()).cpu()                                  # pairwise intersections in one GEMM
    areas = flat.sum(1).cpu()
    order = torch.argsort(scores).tolist()[::-1]
    suppressed = [False] * nums
    keep = []
    for i in range(nums):
        idx = order[i]
        if suppressed[idx]:
            continue
        keep.append(idx)
        for jj in range(i, nums):
            j = order[jj]
            if suppressed[j]:
                continue
            if comput_mmi(float(areas[idx]), float(areas[j]), float(inter[idx, j])) >= thres:
                suppressed[j] = True
    return masks[keep]


def mask_post(inst_masks, thres1=0.63, thres2=0.5, bd_flag=False):
    """Merge masks whose pairwise dice exceeds thres1 (first-come clustering), average each cluster (:403-431)."""
    dice = dice_for(inst_masks).cpu()
    query_num = dice.shape[0]
    taken, clusters = set(), []
    for i in range(query_num):
        if i in taken:
            continue
        members = torch.where(dice[i] > thres1)[0].tolist()
        taken.update(members)
        clusters.append(members)
    merged = []
    for ids in clusters:
        m = inst_masks[ids].mean(dim=0)
        merged.append((m > thres2).float() if bd_flag else m)
    return torch.stack(merged)


class MaskFormer(nn.Module):
    def __init__(self, *, backbone: nn.Module, sem_seg_head: nn.Module, criterion: nn.Module, num_queries: int,
                 object_mask_threshold: float = 0.8, overlap_threshold: float = 0.8, size_divisibility: int = 32,
                 sem_seg_postprocess_before_inference: bool = True, semantic_on: bool = False,
                 instance_on: bool = True, panoptic_on: bool = False, test_topk_per_image: int = 100,
                 test_threshold: float = 0.5, dataset_name: str = "CVPPP"):
        super().__init__()
        self.backbone = backbone
        self.sem_seg_head = sem_seg_head
        self.criterion = criterion
        self.num_queries = num_queries
        self.overlap_threshold = overlap_threshold
        self.object_mask_threshold = object_mask_threshold
        if size_divisibility < 0:
            size_divisibility = getattr(self.backbone, "size_divisibility", 0)
        self.size_divisibility = size_divisibility
        self.sem_seg_postprocess_before_inference = sem_seg_postprocess_before_inference
        self.semantic_on = semantic_on
        self.instance_on = instance_on
        self.panoptic_on = panoptic_on
        self.test_topk_per_image = test_topk_per_image
        self.test_threshold = test_threshold
        if not self.semantic_on:
            assert self.sem_seg_postprocess_before_inference
        self.dataset_name = dataset_name

    @classmethod
    def from_config(cls, cfg, backbone):
        """cfg keys as arch/maskformer.py:72-159; the backbone is passed in (detectron2's builder is not available)."""
        mf = cfg.MODEL.MASK_FORMER
        head = MaskFormerHead(**MaskFormerHead.from_config(cfg, backbone.output_shape()))
        matcher = Point_HungarianMatcher(cost_mask=mf.MASK_WEIGHT, cost_dice=mf.DICE_WEIGHT,
                                         num_points=mf.TRAIN_NUM_POINTS)
        weight_dict = {"loss_mask": mf.MASK_WEIGHT, "loss_dice": mf.DICE_WEIGHT}
        dec_layers = mf.DEC_LAYERS
        if mf.DEEP_SUPERVISION:
            aux = {}
            for i in range(dec_layers - 1):
                aux.update({k + f"_{i}": v for k, v in weight_dict.items()})
            weight_dict.update(aux)
        weight_dict["loss_emb"] = mf.EMB_WEIGHT
        weight_dict["loss_reid_query"] = mf.REID_WEIGHT_QUERY
        weight_dict["loss_reid_query_aux"] = mf.REID_WEIGHT_QUERY * 1.5
        weight_dict["loss_reid_mask"] = mf.REID_WEIGHT_MASK
        weight_dict["loss_refpoints"] = mf.REF_POINTS_WEIGHT
        for i in range(dec_layers - 1):
            if i != 0:
                weight_dict[f"loss_refpoints_{i}"] = mf.REF_POINTS_WEIGHT
        losses = ["masks", "refpoints", "reid_query", "reid_mask"]
        if mf.SEMANTIC_LOSS_ON:
            weight_dict["loss_sem"] = mf.SEM_WEIGHT
            losses.append("sem")
        losses.append("embedding")
        criterion = SetCriterion(head.num_classes, matcher=matcher, weight_dict=weight_dict,
                                 eos_coef=mf.NO_OBJECT_WEIGHT, losses=losses, num_points=mf.TRAIN_NUM_POINTS,
                                 oversample_ratio=mf.OVERSAMPLE_RATIO,
                                 importance_sample_ratio=mf.IMPORTANCE_SAMPLE_RATIO)
        return dict(backbone=backbone, sem_seg_head=head, criterion=criterion, num_queries=mf.NUM_OBJECT_QUERIES,
                    object_mask_threshold=mf.TEST.OBJECT_MASK_THRESHOLD, overlap_threshold=mf.TEST.OVERLAP_THRESHOLD,
                    size_divisibility=mf.SIZE_DIVISIBILITY, sem_seg_postprocess_before_inference=True,
                    semantic_on=mf.TEST.SEMANTIC_ON, instance_on=mf.TEST.INSTANCE_ON, panoptic_on=mf.TEST.PANOPTIC_ON,
                    dataset_name=cfg.DATASET.DATA_TYPE)

    def forward(self, volume, targets=None, train=False):
        features = self.backbone(volume)
        if train:
            outputs, mask_features = self.sem_seg_head(features, targets, criterion=self.criterion)
            losses = self.criterion(outputs, targets, mask_features)
            for k in list(losses.keys()):
                if k in self.criterion.weight_dict:
                    losses[k] *= self.criterion.weight_dict[k]
                else:
                    losses.pop(k)           # not in weight_dict -> not trained on
            return losses

        outputs, _ = self.sem_seg_head(features)
        mask_pred_results = F.interpolate(outputs["pred_masks"], size=(volume.shape[-2], volume.shape[-1]),
                                          mode="bilinear", align_corners=False)
        del outputs
        processed_results: List[torch.Tensor] = []
        processed_boundary_results: List[torch.Tensor] = []
        for mask_pred_result in mask_pred_results:
            if self.instance_on:
                instance_r, boundary_r = self.instance_inference(mask_pred_result)
                processed_results.append(instance_r)
                if boundary_r is not None:
                    processed_boundary_results.append(boundary_r)
        output = torch.cat(processed_results)
        boundary_output = torch.cat(processed_boundary_results) if processed_boundary_results else None
        return output, boundary_output

    def instance_inference(self, mask_pred):
        """[Q, H, W] mask logits at input resolution -> ([1, H, W] int16 instance ids (0 = background), None).
        Thresholds per dataset as the reference: CVPPP 0.69 / merge 0.5, 0.6 + NMS 0.72; BBBC 0.05 / merge 0.15, 0.25."""
        if self.dataset_name == "CVPPP":
            threshold, t1, t2, bd, nms = 0.69, 0.5, 0.6, True, 0.72
        elif self.dataset_name == "BBBC":
            threshold, t1, t2, bd, nms = 0.05, 0.15, 0.25, False, None
        else:
            raise ValueError("instance_inference: unknown dataset %r" % (self.dataset_name,))
        pred_masks = (mask_pred.sigmoid().float() > threshold).float()
        H, W = pred_masks.shape[-2:]
        pred_masks = pred_masks[pred_masks.flatten(1).sum(1) > 40]
        if pred_masks.shape[0] == 0:
            return torch.zeros((1, H, W), dtype=torch.int16, device=mask_pred.device), None
        pred_masks = mask_post(pred_masks, thres1=t1, thres2=t2, bd_flag=bd)
        if nms is not None:
            areas = pred_masks.flatten(1).sum(1)
            pred_masks = mask_nms(pred_masks, (areas / areas.max()).to(pred_masks), thres=nms)
        areas = pred_masks.flatten(1).sum(1)
        pred_masks = pred_masks[torch.argsort(areas)]
        mask_scores = torch.cat([pred_masks.new_zeros((1, H, W)), pred_masks])
        prd_result = torch.argmax(mask_scores, dim=0).to(torch.int16)
        return prd_result[None, :], None
