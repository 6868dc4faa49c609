"""Point-sampled Hungarian matcher.

Mirrors loss/matcher.py:70-222 of the reference (`Point_HungarianMatcher`, `batch_dice_loss`, `batch_sigmoid_ce_loss`):
per image, `num_points` random points are sampled from predicted and target masks, the [Q, G] cost
cost_mask * BCE + cost_dice * dice is formed with two GEMMs on the device, and scipy's linear_sum_assignment solves it.
Same RNG consumption (one torch.rand(1, num_points, 2) per image) so seeded runs match.  The reference's
`torch.cuda.empty_cache()` after every image (matcher.py:160) is dropped: it only stalls the allocator.
"""
import torch
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment
from torch import nn

from .point_features import point_sample


def batch_dice_loss(inputs: torch.Tensor, targets: torch.Tensor):
    inputs = inputs.sigmoid().flatten(1)
    numerator = 2 * torch.einsum("nc,mc->nm", inputs, targets)
    denominator = inputs.sum(-1)[:, None] + targets.sum(-1)[None, :]
    return 1 - (numerator + 1) / (denominator + 1)


def batch_sigmoid_ce_loss(inputs: torch.Tensor, targets: torch.Tensor):
    hw = inputs.shape[1]
    pos = F.binary_cross_entropy_with_logits(inputs, torch.ones_like(inputs), reduction="none")
    neg = F.binary_cross_entropy_with_logits(inputs, torch.zeros_like(inputs), reduction="none")
    loss = torch.einsum("nc,mc->nm", pos, targets) + torch.einsum("nc,mc->nm", neg, (1 - targets))
    return loss / hw


class Point_HungarianMatcher(nn.Module):
    def __init__(self, cost_mask: float = 1, cost_dice: float = 1, num_points: int = 0):
        super().__init__()
        self.cost_mask = cost_mask
        self.cost_dice = cost_dice
        assert cost_mask != 0 or cost_dice != 0, "all costs cant be 0"
        self.num_points = num_points
        self.device_lsap = True            # CUDA inputs: assign on the device; False = scipy on the host (reference)
        self.batch_images = True           # one cost pass for the whole batch when the target masks share a size
        self.pending_status = []

    def _batched_costs(self, outputs, targets):
        """All images of the batch in one pass -- when their target masks share one size: the predicted masks are the
        channels of ONE grid_sample, the targets (padded to the largest instance count) of another, and the costs are
        batched contractions.  Same random points as the per-image loop (one torch.rand(1, P, 2) per image, in order).
        -> (cost [B, Q, Gmax] with zeros past an image's count, counts) or None when the sizes differ."""
        pred = outputs["pred_masks"]
        bs, num_queries = pred.shape[:2]
        counts = [int(t["masks"].shape[0]) for t in targets]
        sizes = {tuple(t["masks"].shape[-2:]) for t in targets}
        if bs == 0 or len(sizes) != 1 or max(counts) == 0:
            return None
        gmax = max(counts)
        coords = torch.cat([torch.rand(1, self.num_points, 2, device=pred.device) for _ in range(bs)])
        if min(counts) == gmax:
            tgt = torch.stack([t["masks"] for t in targets]).to(pred)
        else:
            tgt = pred.new_zeros((bs, gmax) + next(iter(sizes)))
            for b, t in enumerate(targets):
                tgt[b, :counts[b]] = t["masks"]
        tgt_pts = point_sample(tgt, coords, align_corners=False)                # [B, Gmax, P]
        out_pts = point_sample(pred, coords, align_corners=False)               # [B, Q, P]
        with torch.autocast(device_type=pred.device.type, enabled=False):
            out_pts, tgt_pts = out_pts.float(), tgt_pts.float()
            hw = out_pts.shape[2]
            pos = F.binary_cross_entropy_with_logits(out_pts, torch.ones_like(out_pts), reduction="none")
            neg = F.binary_cross_entropy_with_logits(out_pts, torch.zeros_like(out_pts), reduction="none")
            ce = (torch.einsum("bnc,bmc->bnm", pos, tgt_pts) + torch.einsum("bnc,bmc->bnm", neg, (1 - tgt_pts))) / hw
            prob = out_pts.sigmoid()
            numerator = 2 * torch.einsum("bnc,bmc->bnm", prob, tgt_pts)
            denominator = prob.sum(-1)[:, :, None] + tgt_pts.sum(-1)[:, None, :]
            C = self.cost_mask * ce + self.cost_dice * (1 - (numerator + 1) / (denominator + 1))
        return C, counts

    @torch.no_grad()
    def memory_efficient_forward(self, outputs, targets):
        bs, num_queries = outputs["pred_masks"].shape[:2]
        if self.batch_images:
            got = self._batched_costs(outputs, targets)
            if got is not None:
                C, counts = got
                if (self.device_lsap and C.is_cuda and max(counts) <= num_queries and num_queries <= 1024
                        and max(counts) <= 512):
                    return self._assign_padded(C, counts)
                C = C.cpu()
                indices = [linear_sum_assignment(C[b, :, :g]) for b, g in enumerate(counts)]
                return [(torch.as_tensor(i, dtype=torch.int64), torch.as_tensor(j, dtype=torch.int64))
                        for i, j in indices]
        costs = []
        for b in range(bs):
            out_mask = outputs["pred_masks"][b][:, None]                      # [Q, 1, H, W]
            tgt_mask = targets[b]["masks"].to(out_mask)[:, None]              # [G, 1, Ht, Wt]
            point_coords = torch.rand(1, self.num_points, 2, device=out_mask.device)
            tgt_pts = point_sample(tgt_mask, point_coords.repeat(tgt_mask.shape[0], 1, 1),
                                   align_corners=False).squeeze(1)
            out_pts = point_sample(out_mask, point_coords.repeat(out_mask.shape[0], 1, 1),
                                   align_corners=False).squeeze(1)
            with torch.autocast(device_type=out_mask.device.type, enabled=False):
                out_pts, tgt_pts = out_pts.float(), tgt_pts.float()
                C = self.cost_mask * batch_sigmoid_ce_loss(out_pts, tgt_pts) \
                    + self.cost_dice * batch_dice_loss(out_pts, tgt_pts)
            costs.append(C.reshape(num_queries, -1))
        counts = [int(C.shape[1]) for C in costs]
        if (self.device_lsap and bs > 0 and costs[0].is_cuda and max(counts) <= num_queries
                and num_queries <= 1024 and max(counts) <= 512):
            return self._assign_on_device(costs, counts, num_queries)
        # one device -> host transfer per call instead of one per image
        indices = [linear_sum_assignment(C.cpu()) for C in costs]
        return [(torch.as_tensor(i, dtype=torch.int64), torch.as_tensor(j, dtype=torch.int64)) for i, j in indices]

    def _assign_on_device(self, costs, counts, num_queries):
        """Hungarian matching with the HIP kernel (csrc/lsap.hip): no host synchronisation, the index tensors stay on the
        device (the reference moves every cost matrix to the host and runs scipy, matcher.py:154-165).  Infeasible
        problems (NaN / inf costs, where scipy raises) are flagged in `self.pending_status`; `check()` raises for them."""
        from .. import fused_ops
        dev = costs[0].device
        gmax = max(1, max(counts))
        padded = torch.zeros((len(costs), num_queries, gmax), dtype=torch.float32, device=dev)
        for b, C in enumerate(costs):
            if counts[b]:
                padded[b, :, :counts[b]] = C
        return self._assign_padded(padded, counts)

    def _assign_padded(self, padded, counts):
        from .. import fused_ops
        dev = padded.device
        rows, status = fused_ops.lsap(padded, torch.tensor(counts, dtype=torch.int32).to(dev, non_blocking=True))
        self.pending_status.append(status)
        if len(self.pending_status) > 256:             # nobody called check(): fold the backlog into one flag tensor
            self.pending_status = [torch.cat(self.pending_status).max().reshape(1)]
        # scipy lists the pairs by ascending prediction index: one sort for all problems (unmatched columns hold -1 and are
        # moved behind every query index first; a problem's matched queries are distinct, so its first g entries are the
        # per-problem sort)
        vals, order = torch.sort(torch.where(rows < 0, padded.shape[1], rows).long(), dim=1)
        return [(vals[b, :g], order[b, :g]) for b, g in enumerate(counts)]

    @torch.no_grad()
    def forward_many(self, outputs_list, targets):
        """The matchings of several predictions against the same targets -- the decoder's ten prediction heads -- as one
        call: the cost matrices are formed prediction by prediction exactly as `forward` forms them (same random points
        in the same order), all assignment problems then go to the device in ONE launch (a 300 x 60 problem keeps one
        workgroup busy for about a millisecond: ten launches of two problems each left the other 254 compute units idle
        for 10 ms of every BASELINE configs[3] training step).  -> [forward(o, targets) for o in outputs_list]."""
        if not outputs_list:
            return []
        pred = outputs_list[0]["pred_masks"]
        bs, num_queries = pred.shape[:2]
        counts = [int(t["masks"].shape[0]) for t in targets]
        sizes = {tuple(t["masks"].shape[-2:]) for t in targets}
        same = all(o["pred_masks"].shape[:2] == pred.shape[:2] and o["pred_masks"].is_cuda for o in outputs_list)
        if not (self.batch_images and self.device_lsap and same and bs > 0 and len(sizes) == 1 and 0 < max(counts) <= min(num_queries, 512)
                and num_queries <= 1024):
            return [self.memory_efficient_forward(o, targets) for o in outputs_list]
        costs = [self._batched_costs(o, targets)[0] for o in outputs_list]
        flat = self._assign_padded(torch.cat(costs), counts * len(outputs_list))
        return [flat[k * bs:(k + 1) * bs] for k in range(len(outputs_list))]

    def check(self):
        """Raise if any assignment since the last call was infeasible (one host synchronisation)."""
        pending, self.pending_status = self.pending_status, []
        if pending and bool(torch.cat(pending).any()):
            raise ValueError("cost matrix is infeasible (NaN or infinite costs)")

    @torch.no_grad()
    def forward(self, outputs, targets):
        """outputs["pred_masks"] [B, Q, H, W]; targets: list of dicts with "masks" [G_b, Ht, Wt].
        -> list of (pred_idx, tgt_idx) int64 tensors, len = min(Q, G_b)."""
        return self.memory_efficient_forward(outputs, targets)

    def __repr__(self, _repr_indent=4):
        head = "Matcher " + self.__class__.__name__
        body = ["cost_mask: {}".format(self.cost_mask), "cost_dice: {}".format(self.cost_dice)]
        return "\n".join([head] + [" " * _repr_indent + line for line in body])
