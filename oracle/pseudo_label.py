"""Oracle for the pseudo-label filter (TEST INFRASTRUCTURE).

Restates the per-batch tensor code of /root/reference/gen_data.py:132-164 and the --topk
post-filter of :196-215 with torch on the CPU.  Pinned by tests/golden/pseudo_label.npz: that code is
inline in gen_data.py's main(), so tools/make_golden_pseudo.py runs main() itself end to end with
in-memory stand-ins for what surrounds it (clip, nerv, build_model, build_dataset; a fake classifier returns
prescribed probabilities) and reads the per-sample pseudo-labels off the symlink tree it writes, for 30
combinations of --tta / --tta_consistent / --tta_min_prob / --conf_thresh / --topk.
"""
import torch


def select(pred_probs, conf_thresh, tta=False, tta_consistent=False, tta_min_prob=False):
    pred_probs = pred_probs.float()
    if tta:
        probs = pred_probs.unflatten(0, (-1, 4))                          # :136
        tta_mask = torch.ones(probs.shape[0], dtype=torch.bool)           # :137
        if tta_consistent:                                                # :139-143
            pred_cls = probs.argmax(dim=-1)
            tta_mask &= (pred_cls[:, 0] == pred_cls[:, 1]) & (pred_cls[:, 0] == pred_cls[:, 2]) & \
                (pred_cls[:, 0] == pred_cls[:, 3])
        if tta_min_prob:                                                  # :145-147
            min_probs = probs.max(-1).values.min(-1).values
            tta_mask &= (min_probs > conf_thresh)
        probs = probs.mean(dim=1)                                         # :148
    else:
        probs = pred_probs                                                # :150
    max_probs, pred_labels = probs.max(dim=-1)                            # :155
    sel_mask = (max_probs > conf_thresh)                                  # :156
    if tta:
        sel_mask &= tta_mask                                              # :157-158
    return dict(probs=probs, pred=pred_labels, max_prob=max_probs, selected=sel_mask)


def topk_per_class(pred, max_prob, selected, n_classes, topk):
    """:196-215: per predicted class, the topk most confident selected samples."""
    keep = torch.zeros_like(selected)
    for c in range(n_classes):
        members = [i for i in range(len(pred)) if selected[i] and int(pred[i]) == c]
        if not members:
            continue
        probs = torch.tensor([float(max_prob[i]) for i in members])
        k = min(topk, probs.shape[0])
        for j in probs.topk(k).indices.tolist():
            keep[members[j]] = True
    return keep
