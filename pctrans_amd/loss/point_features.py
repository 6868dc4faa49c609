"""Point sampling helpers (detectron2's PointRend `point_features`, which the reference imports but does not vendor:
loss/matcher.py:13, loss/maskformer_criterion.py:13-16).  Restated from their published definitions."""
import torch
from torch.nn import functional as F


def point_sample(input, point_coords, **kwargs):
    """Bilinear samples of `input` [N, C, H, W] at `point_coords` [N, P, 2] (or [N, Hg, Wg, 2]) in [0, 1] x [0, 1]
    (x, y), i.e. F.grid_sample on 2 * coords - 1."""
    add_dim = point_coords.dim() == 3
    if add_dim:
        point_coords = point_coords.unsqueeze(2)
    output = F.grid_sample(input, 2.0 * point_coords - 1.0, **kwargs)
    return output.squeeze(3) if add_dim else output


def get_uncertain_point_coords_with_randomness(coarse_logits, uncertainty_func, num_points, oversample_ratio,
                                               importance_sample_ratio):
    """PointRend importance sampling: draw oversample_ratio * num_points uniform points, keep the
    importance_sample_ratio * num_points most uncertain ones, fill the rest with fresh uniform points."""
    assert oversample_ratio >= 1 and 0 <= importance_sample_ratio <= 1
    num_boxes = coarse_logits.shape[0]
    num_sampled = int(num_points * oversample_ratio)
    point_coords = torch.rand(num_boxes, num_sampled, 2, device=coarse_logits.device)
    point_logits = point_sample(coarse_logits, point_coords, align_corners=False)
    point_uncertainties = uncertainty_func(point_logits)
    num_uncertain_points = int(importance_sample_ratio * num_points)
    num_random_points = num_points - num_uncertain_points
    idx = torch.topk(point_uncertainties[:, 0, :], k=num_uncertain_points, dim=1)[1]
    shift = num_sampled * torch.arange(num_boxes, dtype=torch.long, device=coarse_logits.device)
    idx = idx + shift[:, None]
    point_coords = point_coords.view(-1, 2)[idx.view(-1), :].view(num_boxes, num_uncertain_points, 2)
    if num_random_points > 0:
        point_coords = torch.cat(
            [point_coords, torch.rand(num_boxes, num_random_points, 2, device=coarse_logits.device)], dim=1)
    return point_coords
