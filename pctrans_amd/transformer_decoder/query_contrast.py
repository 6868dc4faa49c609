"""Query-contrast pair selection (training only; last decoder layer).

Restates select_pos_neg_query / select_pos_neg_mask / dice_for of the reference
(transformer_decoder/mask2former_transformer_decoder.py:800-927): every unmatched query is assigned to the matched
("positive") query it is most cosine-similar to; for each matched query with a non-empty cluster the cluster members
are positives and all other queries negatives.  Outputs keep the reference's structure (lists of dicts consumed by
SetCriterion.loss_reid_query / loss_reid_mask, loss/maskformer_criterion.py of the reference):
    query items: {'contrast' [n_pos+n_neg, 1], 'label', 'aux_consin' [n_pos+n_sample, 1], 'aux_label'}
    mask  items: {'contrast' [n_pos+n_neg, 1] (pairwise soft dice of sigmoid masks), 'label'}
Index bookkeeping is done once per image on the host (`.tolist()` on the matcher's indices, as the reference does);
negative sub-sampling uses `random.sample` exactly like the reference so seeded runs draw the same negatives.
"""
import random

import torch
from torch.nn import functional as F


def dice_for(inputs):
    """Pairwise soft dice between the sigmoid masks of one image: [Q, ...] -> [Q, Q] (:917-927)."""
    x = inputs.flatten(1).sigmoid()
    numerator = x @ x.transpose(-2, -1)
    s = x.sum(-1)
    return (2 * numerator + 1) / (s[:, None] + s[None, :] + 1)


def _clusters(emb_dist_b, pos_ids, query_num):
    """pos_ids: matched query ids (list).  -> per positive id: list of unmatched ids whose most similar positive it is."""
    rest_ids = [i for i in range(query_num) if i not in set(pos_ids)]
    if not rest_ids or not pos_ids:
        return [[] for _ in pos_ids]
    sub = emb_dist_b[rest_ids][:, pos_ids]
    nearest = torch.argmax(sub, dim=1).tolist()
    nearest_pos = [pos_ids[i] for i in nearest]
    return [[r for r, np_ in zip(rest_ids, nearest_pos) if np_ == pid] for pid in pos_ids]


def select_pos_neg_query(query, emb_dist, pos_indices):
    """query [Q, N, C]; emb_dist [N, Q, Q] cosine similarities; pos_indices: matcher output [(src_idx, tgt_idx)] per image."""
    query = query.transpose(0, 1)
    bz, query_num = query.shape[0], query.shape[1]
    one, zero = query.new_tensor(1), query.new_tensor(0)
    items = []
    for b in range(bz):
        pos_ids = pos_indices[b][0].tolist()
        for pos_id, cluster in zip(pos_ids, _clusters(emb_dist[b], pos_ids, query_num)):
            if not cluster:
                continue
            key = query[b][pos_id].unsqueeze(0)
            members = set(cluster) | {pos_id}
            neg_ids = [i for i in range(query_num) if i not in members]
            pos_embed, neg_embed = query[b][cluster], query[b][neg_ids]
            contrastive_embed = torch.cat([pos_embed, neg_embed], dim=0)
            label = torch.cat([one.repeat(len(pos_embed)), zero.repeat(len(neg_embed))], dim=0)
            contrast = torch.einsum("nc,kc->nk", contrastive_embed, key)
            num_sample_neg = len(neg_embed) if len(pos_embed) * 10 >= len(neg_embed) else len(pos_embed) * 10
            sample_ids = random.sample(list(range(len(neg_embed))), num_sample_neg)
            aux_embed = torch.cat([pos_embed, neg_embed[sample_ids]], dim=0)
            aux_label = torch.cat([one.repeat(len(pos_embed)), zero.repeat(num_sample_neg)], dim=0)
            cosine = torch.einsum("nc,kc->nk", F.normalize(aux_embed.float(), dim=1), F.normalize(key.float(), dim=1))
            items.append({"contrast": contrast, "label": label, "aux_consin": cosine, "aux_label": aux_label})
    return items


def select_pos_neg_mask(query_mask, emb_dist, pos_indices):
    """query_mask [N, Q, H, W] mask logits."""
    bz, query_num = query_mask.shape[0], query_mask.shape[1]
    one, zero = query_mask.new_tensor(1), query_mask.new_tensor(0)
    items = []
    for b in range(bz):
        pos_ids = pos_indices[b][0].tolist()
        clusters = _clusters(emb_dist[b], pos_ids, query_num)
        if not any(clusters):
            continue
        dice_query = dice_for(query_mask[b])
        for pos_id, cluster in zip(pos_ids, clusters):
            if not cluster:
                continue
            members = set(cluster) | {pos_id}
            neg_ids = [i for i in range(query_num) if i not in members]
            label = torch.cat([one.repeat(len(cluster)), zero.repeat(len(neg_ids))], dim=0)
            contrast = torch.cat([dice_query[pos_id][cluster][:, None], dice_query[pos_id][neg_ids][:, None]])
            items.append({"contrast": contrast, "label": label})
    return items


def query_contrast_items(output, outputs_mask, indices):
    """Last-layer hook of the decoder (:618-622): output [Q, N, C] query embeddings, outputs_mask [N, Q, H, W]."""
    q = output.permute(1, 0, 2)
    emb_dist = F.cosine_similarity(q.unsqueeze(2), q.unsqueeze(1), dim=-1)          # [N, Q, Q]
    return select_pos_neg_query(output, emb_dist, indices), select_pos_neg_mask(outputs_mask, emb_dist, indices)
