"""Oracle for logits / view aggregation (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates /root/reference/models/clip_cls.py in plain torch fp32:
ZSCLIPClassifier.forward (:131-162), _aggregate_logits (:104-121),
_aggregate_probs (:123-129), and the tail of FSCLIPClassifier.forward (:319-343).
Pinned against the reference's own classes (imported with in-memory stubs of
`clip` / `nerv`) by tools/make_golden_models.py -> tests/golden/classify_*.npz.
"""
import torch
import torch.nn.functional as F


def aggregate_logits(logits, valid_masks, agg_func):
    """clip_cls.py:104-121.  logits [B, T, K], valid_masks [B, T] bool."""
    if agg_func == 'sum':
        return logits.sum(1)
    if agg_func == 'mean':
        return logits.sum(1) / valid_masks.float().sum(1, keepdim=True)
    if agg_func == 'max':
        # the reference subtracts the [B, T] mask without unsqueezing (clip_cls.py:117),
        # which cannot broadcast against [B, T, K] and raises; this is the evident intent
        logits = logits - (1. - valid_masks.float())[..., None] * 1e6
        return logits.max(1)[0]
    raise NotImplementedError(agg_func)


def aggregate_probs(logits, valid_masks):
    """clip_cls.py:123-129: per-view softmax, mask, mean over valid views."""
    vm = valid_masks.float()
    probs = logits.softmax(dim=-1) * vm[..., None]
    return probs.sum(1) / vm.sum(1, keepdim=True)


def zs_forward(img_feats, valid_masks, text_feats, logit_scale, agg_func):
    """clip_cls.py:148-161.  img_feats [Nv, C] for the valid views in row-major (b, t)
    order (what `imgs[valid_masks]` feeds encode_image), NOT normalised (:148);
    text_feats [K, C] already L2-normalised (:85)."""
    B, T = valid_masks.shape
    logits = logit_scale * img_feats @ text_feats.T
    full = torch.zeros(B, T, text_feats.shape[0]).type_as(logits)
    full[valid_masks] = logits
    return dict(full_logits=full, valid_masks=valid_masks,
                logits=aggregate_logits(full, valid_masks, agg_func),
                probs=aggregate_probs(full, valid_masks))


def fs_tail(full_img_feats, valid_masks, text_feats, logit_scale, agg_func):
    """clip_cls.py:326-343 after the adapter: normalise, mask, logits, aggregate.
    full_img_feats [B, T, C]; text_feats [K, C] normalised."""
    f = F.normalize(full_img_feats, p=2, dim=-1)
    f = f * valid_masks.float().unsqueeze(-1)
    full = logit_scale * f @ text_feats.T
    return dict(full_logits=full, valid_masks=valid_masks,
                logits=aggregate_logits(full, valid_masks, agg_func),
                probs=aggregate_probs(full, valid_masks))
