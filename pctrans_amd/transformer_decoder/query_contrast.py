"""Query-contrast pair selection (training only; last decoder layer).

Restates select_pos_neg_query / select_pos_neg_mask / dice_for of the reference
(transformer_decoder/mask2former_transformer_decoder.py:800-927): every unmatched query is assigned to the matched
("positive") query it is most cosine-similar to; for each matched query with a non-empty cluster the cluster members
are positives and all other queries negatives.  Outputs keep the reference's structure (lists of dicts consumed by
SetCriterion.loss_reid_query / loss_reid_mask, loss/maskformer_criterion.py of the reference):
    query items: {'contrast' [n_pos+n_neg, 1], 'label', 'aux_consin' [n_pos+n_sample, 1], 'aux_label'}
    mask  items: {'contrast' [n_pos+n_neg, 1] (pairwise soft dice of sigmoid masks), 'label'}

The reference builds every item with its own gathers, concatenations and einsums (a dozen launches per matched query and
loss, thousands per step).  Here the bookkeeping -- which query is whose positive / negative, and the `random.sample`
draws of the auxiliary negatives, in the reference's order so seeded runs pick the same ones -- is plain Python on ONE
device->host copy of the nearest-positive table, and the numbers of ALL items come from a handful of batched gathers:
an item's positives and negatives are always "every other query", so all items share the shape [Q - 1].  The dicts hold
views of those batched tensors; the returned lists also carry them whole (`ContrastItems.batched`) so that the criterion
reduces them without a per-item loop.
"""
import random

import torch
from torch.nn import functional as F


class ContrastItems(list):
    """The reference's list of per-item dicts, plus `.batched`: the same numbers as dense tensors (None for a plain list)."""
    batched = None


def dice_for(inputs):
    """Pairwise soft dice between the sigmoid masks of one image: [Q, ...] -> [Q, Q] (:917-927)."""
    x = inputs.flatten(1).sigmoid()
    numerator = x @ x.transpose(-2, -1)
    s = x.sum(-1)
    return (2 * numerator + 1) / (s[:, None] + s[None, :] + 1)


def _plan(emb_dist, pos_indices):
    """Host-side bookkeeping for the whole batch.  emb_dist [N, Q, Q]; pos_indices: matcher output per image.
    -> list of (image, positive id, cluster ids, negative ids), in the reference's order (image, then matched query)."""
    bz, query_num = emb_dist.shape[0], emb_dist.shape[1]
    lens = [int(p[0].numel()) for p in pos_indices]
    flat = torch.cat([p[0].reshape(-1) for p in pos_indices]).tolist() if sum(lens) else []
    pos_lists, at = [], 0
    for n in lens:
        pos_lists.append(flat[at:at + n])
        at += n
    pmax = max(lens) if lens else 0
    if pmax == 0:
        return []
    # nearest positive of every query: argmax over the image's positive columns in matcher order (first maximum wins, as
    # torch.argmax on the reference's emb_dist[rest][:, pos] sub-matrix)
    pad = torch.tensor([pl + [0] * (pmax - len(pl)) for pl in pos_lists], dtype=torch.long)
    valid = torch.tensor([[True] * len(pl) + [False] * (pmax - len(pl)) for pl in pos_lists])
    pad, valid = pad.to(emb_dist.device), valid.to(emb_dist.device)
    sub = emb_dist.gather(2, pad[:, None, :].expand(bz, query_num, pmax))
    nearest = sub.masked_fill(~valid[:, None, :], float("-inf")).argmax(-1).tolist()          # the one device->host copy
    plan = []
    for b in range(bz):
        pos_ids = pos_lists[b]
        if not pos_ids:
            continue
        pos_set = set(pos_ids)
        clusters = [[] for _ in pos_ids]
        for r in range(query_num):
            if r not in pos_set:
                clusters[nearest[b][r]].append(r)
        for pos_id, cluster in zip(pos_ids, clusters):
            if not cluster:
                continue
            members = set(cluster) | {pos_id}
            plan.append((b, pos_id, cluster, [i for i in range(query_num) if i not in members]))
    return plan


def _index_tensors(plan, device):
    img = torch.tensor([p[0] for p in plan], dtype=torch.long)
    key = torch.tensor([p[1] for p in plan], dtype=torch.long)
    order = torch.tensor([p[2] + p[3] for p in plan], dtype=torch.long)                         # [I, Q - 1]
    n_pos = torch.tensor([len(p[2]) for p in plan], dtype=torch.long)
    return img.to(device), key.to(device), order.to(device), n_pos.to(device)


def select_pos_neg_query(query, emb_dist, pos_indices, _plan_cache=None):
    """query [Q, N, C]; emb_dist [N, Q, Q] cosine similarities; pos_indices: matcher output [(src_idx, tgt_idx)] per image."""
    query = query.transpose(0, 1)                                                               # [N, Q, C]
    plan = _plan(emb_dist, pos_indices) if _plan_cache is None else _plan_cache
    items = ContrastItems()
    if not plan:
        return items
    dev = query.device
    img, key, order, n_pos = _index_tensors(plan, dev)
    n_other = order.shape[1]
    keys = query[img, key]                                                                      # [I, C]
    contrast = torch.einsum("iqc,ic->iq", query[img[:, None], order], keys)                     # [I, Q - 1]
    label = (torch.arange(n_other, device=dev)[None, :] < n_pos[:, None]).to(query.dtype)
    # auxiliary cosine pairs: the cluster plus at most 10 negatives per positive, drawn like the reference (:856-858)
    rows_img, rows_q, rows_item, rows_label, pad_idx, counts = [], [], [], [], [], []
    for i, (b, _, cluster, neg_ids) in enumerate(plan):
        num_sample_neg = len(neg_ids) if len(cluster) * 10 >= len(neg_ids) else len(cluster) * 10
        sample_ids = random.sample(list(range(len(neg_ids))), num_sample_neg)
        qs = cluster + [neg_ids[s] for s in sample_ids]
        pad_idx.append(list(range(len(rows_q), len(rows_q) + len(qs))))
        rows_img += [b] * len(qs)
        rows_q += qs
        rows_item += [i] * len(qs)
        rows_label += [1.0] * len(cluster) + [0.0] * num_sample_neg
        counts.append(len(qs))
    unit = F.normalize(query.float(), dim=-1)
    rows_img_t = torch.tensor(rows_img, dtype=torch.long).to(dev)
    rows_q_t = torch.tensor(rows_q, dtype=torch.long).to(dev)
    rows_item_t = torch.tensor(rows_item, dtype=torch.long).to(dev)
    cosine = (unit[rows_img_t, rows_q_t] * unit[img, key][rows_item_t]).sum(-1)                 # [T]
    aux_label = torch.tensor(rows_label, dtype=query.dtype).to(dev)
    tmax = max(counts)
    pad = torch.tensor([p + [0] * (tmax - len(p)) for p in pad_idx], dtype=torch.long).to(dev)  # rows of item i, padded
    pad_valid = torch.tensor([[1.0] * len(p) + [0.0] * (tmax - len(p)) for p in pad_idx]).to(dev)
    items.batched = {"contrast": contrast, "label": label, "n_pos": n_pos, "aux_consin": cosine, "aux_label": aux_label,
                     "aux_rows": pad, "aux_valid": pad_valid, "aux_count": torch.tensor(counts, dtype=torch.float32).to(dev)}
    at = 0
    for i, c in enumerate(counts):
        items.append({"contrast": contrast[i].unsqueeze(1), "label": label[i],
                      "aux_consin": cosine[at:at + c].unsqueeze(1), "aux_label": aux_label[at:at + c]})
        at += c
    return items


def select_pos_neg_mask(query_mask, emb_dist, pos_indices, _plan_cache=None):
    """query_mask [N, Q, H, W] mask logits."""
    plan = _plan(emb_dist, pos_indices) if _plan_cache is None else _plan_cache
    items = ContrastItems()
    if not plan:
        return items
    dev = query_mask.device
    img, key, order, n_pos = _index_tensors(plan, dev)
    x = query_mask.flatten(2).sigmoid()                                                         # dice_for, all images at once
    s = x.sum(-1)
    dice = (2 * torch.bmm(x, x.transpose(1, 2)) + 1) / (s[:, :, None] + s[:, None, :] + 1)     # [N, Q, Q]
    contrast = dice[img[:, None], key[:, None], order]                                          # [I, Q - 1]
    label = (torch.arange(order.shape[1], device=dev)[None, :] < n_pos[:, None]).to(query_mask.dtype)
    items.batched = {"contrast": contrast, "label": label, "n_pos": n_pos}
    for i in range(len(plan)):
        items.append({"contrast": contrast[i].unsqueeze(1), "label": label[i]})
    return items


def query_contrast_items(output, outputs_mask, indices):
    """Last-layer hook of the decoder (:618-622): output [Q, N, C] query embeddings, outputs_mask [N, Q, H, W]."""
    q = output.permute(1, 0, 2)
    emb_dist = F.cosine_similarity(q.unsqueeze(2), q.unsqueeze(1), dim=-1)          # [N, Q, Q]
    plan = _plan(emb_dist, indices)
    return (select_pos_neg_query(output, emb_dist, indices, _plan_cache=plan),
            select_pos_neg_mask(outputs_mask, emb_dist, indices, _plan_cache=plan))
