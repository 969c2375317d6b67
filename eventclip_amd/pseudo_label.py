"""Pseudo-label selection for self-training (the reference's gen_data.py).

``select`` is the per-batch tensor logic of gen_data.py:132-164 as one HIP kernel
(``ec_pseudo_label``); ``topk_per_class`` is the ``--topk`` post-filter of gen_data.py:196-215;
``PseudoLabeler`` strings them together with the classifier and the four TTA views of
``Event2ImagePipeline.tta`` so that labelling a training set is one pass over raw events.
"""
import torch

from . import _lib


def select(probs, conf_thresh, tta=False, tta_consistent=False, tta_min_prob=False, views=4):
    """probs: CUDA float [B * views, K] when ``tta`` (view-minor, as ``img.flatten(0, 1)`` of the
    reference's [B, 4, ...] batches orders them, gen_data.py:126-127,136) else [B, K].

    Returns dict(probs [B, K] view-mean, pred int64 [B], max_prob float [B], selected bool [B])."""
    dev = _lib.require_gpu()
    V = views if tta else 1
    p = probs.float().contiguous()
    assert p.is_cuda and p.dim() == 2 and p.shape[0] % V == 0
    B, K = p.shape[0] // V, p.shape[1]
    mean = torch.empty((B, K), dtype=torch.float32, device=dev)
    pred = torch.empty((B,), dtype=torch.int32, device=dev)
    mx = torch.empty((B,), dtype=torch.float32, device=dev)
    sel = torch.empty((B,), dtype=torch.uint8, device=dev)
    rc = _lib.lib().ec_pseudo_label(_lib.ptr(p), B, V, K, float(conf_thresh), int(bool(tta_consistent)),
                                    int(bool(tta_min_prob)), _lib.ptr(mean), _lib.ptr(pred),
                                    _lib.ptr(mx), _lib.ptr(sel), _lib.stream_ptr())
    _lib.check(rc, 'ec_pseudo_label')
    return dict(probs=mean, pred=pred.long(), max_prob=mx, selected=sel.bool())


def topk_per_class(pred, max_prob, selected, n_classes, topk):
    """gen_data.py:196-215: among the selected samples predicted as class c keep the ``topk`` most
    confident ones.  Returns a bool mask [B] (all selected samples when ``topk <= 0``)."""
    if topk <= 0:
        return selected.clone()
    keep = torch.zeros_like(selected)
    idx = torch.nonzero(selected).flatten()
    for c in torch.unique(pred[idx]).tolist():
        members = idx[pred[idx] == c]
        k = min(int(topk), int(members.numel()))
        keep[members[max_prob[members].topk(k).indices]] = True
    return keep


class PseudoLabeler:
    """classifier: ZSCLIPClassifier / FSCLIPClassifier; pipeline: Event2ImagePipeline."""

    def __init__(self, classifier, pipeline, conf_thresh, tta=False, tta_consistent=False,
                 tta_min_prob=False):
        self.classifier, self.pipeline = classifier, pipeline
        self.conf_thresh, self.tta = float(conf_thresh), bool(tta)
        self.tta_consistent, self.tta_min_prob = bool(tta_consistent), bool(tta_min_prob)

    @torch.no_grad()
    def __call__(self, events, n_events=None, center=False):
        """Raw events of a batch -> the ``select`` dict (labels for the samples with ``selected``)."""
        if self.tta:
            if center:   # centre once, in place, then flip (caltech.py:176 precedes event2img.py:97-103)
                events, n_events = self._centered(events, n_events)
            views = [self.classifier(b)['probs'] for b in self.pipeline.tta(events, n_events)]
            probs = torch.stack(views, 1).flatten(0, 1)                  # [B * 4, K], view-minor
        else:
            probs = self.classifier(self.pipeline(events, n_events, center=center))['probs']
        return select(probs, self.conf_thresh, self.tta, self.tta_consistent, self.tta_min_prob)

    def _centered(self, events, n_events):
        import numpy as np
        from . import vis
        dev = _lib.require_gpu()
        if isinstance(events, (list, tuple)):
            events, n_events = self.pipeline._concat(events, dev)
        offs = np.concatenate([[0], np.cumsum(n_events)])
        sr = torch.tensor(np.stack([offs[:-1], offs[1:]], 1), dtype=torch.int64, device=dev)
        vis.center_events_device(events, sr, self.pipeline.resolution)
        return events, n_events
