"""``FTCLIPClassifier`` of the reference (models/clip_cls_ft.py) for SERVING fine-tuned checkpoints.

The reference fine-tunes CLIP's vision tower (all of it, sub-sets, or LoRA factors injected into
every attention block, clip_cls_ft.py:44-80, lora.py:384-403) and saves ``model.visual.*`` next to
``text_feats`` / ``adapter.*`` (clip_cls_ft.py:313-321).  Its forward (:196-243) is the few-shot forward
with the identity adapter: encode the valid views, L2-normalise, logits against the (learned) text
features, aggregate.  This class is that forward on the HIP path plus the checkpoint format:
``load_state_dict`` takes a reference checkpoint as it is -- LoRA-injected keys included, folded into
plain weights once (eventclip_amd.lora) -- and repacks the tower; ``state_dict`` emits the reference's
key set.  TRAINING through the towers (``calc_train_loss`` + backward) is not built (out
of scope): ``train(True)`` raises.
"""
import torch

from . import lora as elora
from .clip_cls import FSCLIPClassifier

FT_ADAPTER_DEFAULTS = dict(adapter_type='text-identity', residual=True)
FT_LOSS_DEFAULTS = dict(use_logits_loss=True, use_probs_loss=False)


class FTCLIPClassifier(FSCLIPClassifier):
    """Fine-tuned CLIP for few-shot classification, inference only (clip_cls_ft.py:15-333)."""

    def __init__(self, adapter_dict=None, clip_dict=None, loss_dict=None):
        ad = dict(FT_ADAPTER_DEFAULTS if adapter_dict is None else adapter_dict)
        kind = ad['adapter_type'].lower()
        if (kind[len('text-'):] if kind.startswith('text-') else kind) != 'identity':
            raise AssertionError('FTCLIPClassifier only supports the identity adapter (clip_cls_ft.py:119)')
        super().__init__(adapter_dict=ad, clip_dict=clip_dict,
                         loss_dict=dict(FT_LOSS_DEFAULTS) if loss_dict is None else loss_dict)
        # which parts of the tower were trained (lora / only_conv1 / only_bias / ..., clip_cls_ft.py:50-80)
        # is a property of the checkpoint being served; kept for API parity
        self.lora = self.clip_dict.get('lora', -1)

    # ---- checkpoints: model.visual.* travels with the classifier (clip_cls_ft.py:313-333) ----
    def state_dict(self, *args, **kwargs):
        w = super(FSCLIPClassifier, self).state_dict(*args, **kwargs)    # the ZS filter drops model.*
        vis = {'model.visual.' + k: v for k, v in self.model.visual.state_dict().items()}
        return {**vis, **w}

    def load_state_dict(self, state_dict, strict=True):
        sd = dict(state_dict)
        vis = {k[len('model.visual.'):]: v for k, v in sd.items() if k.startswith('model.visual.')}
        rest = {k: v for k, v in sd.items() if not k.startswith('model.')}
        if vis:
            merged = elora.merge_lora_visual(vis)        # plain checkpoints pass through unchanged
            clip_sd = self.model.state_dict()
            new = {k: v for k, v in clip_sd.items() if not k.startswith('visual.')}
            missing = [k for k in clip_sd if k.startswith('visual.') and k[len('visual.'):] not in merged]
            if missing and strict:
                raise KeyError(f'fine-tuned checkpoint lacks {missing[:3]} ...')
            for k, v in clip_sd.items():
                if k.startswith('visual.'):
                    new[k] = torch.as_tensor(merged.get(k[len('visual.'):], v)).to(v.dtype)
            self.model.load_state_dict(new, strict=strict)               # repacks the 16-bit copies lazily
        elif strict:
            raise KeyError('no model.visual.* entries: not an FTCLIPClassifier checkpoint')
        return super().load_state_dict(rest, strict=strict)

    def train(self, mode=True):
        if mode:
            raise NotImplementedError(
                'fine-tuning through the CLIP towers is not built on the MI355X path (DESIGN.md, out of '
                'scope); train with the reference and serve the checkpoint here')
        return super().train(False)
