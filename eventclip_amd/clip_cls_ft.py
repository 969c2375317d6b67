"""``FTCLIPClassifier`` of the reference (models/clip_cls_ft.py): serving fine-tuned checkpoints, and the model
side of fine-tuning (the optimisation step itself is ``eventclip_amd.ft.FTTrainer``).

The reference fine-tunes CLIP's vision tower (all of it, sub-sets, or LoRA factors injected into
every attention block, clip_cls_ft.py:44-80, lora.py:384-403) and saves ``model.visual.*`` next to
``text_feats`` / ``adapter.*`` (clip_cls_ft.py:313-321).  Its forward (:196-243) is the few-shot forward
with the identity adapter: encode the valid views, L2-normalise, logits against the (learned) text
features, aggregate.  This class is that forward on the HIP path plus the checkpoint format:
``load_state_dict`` takes a reference checkpoint as it is -- LoRA-injected keys included, folded into
plain weights once (eventclip_amd.lora) -- and repacks the tower; ``state_dict`` emits the reference's
key set.  With an ``FTTrainer`` attached the classifier is in the middle of a fine-tuning run: ``forward``
then encodes through the trainer's (LoRA-merged) operand copies and ``state_dict`` writes the trainer's
``model.visual.*`` entries, LoRA factors under the reference's key names included.
"""
import torch

from . import lora as elora
from .clip_cls import FSCLIPClassifier

FT_ADAPTER_DEFAULTS = dict(adapter_type='text-identity', residual=True)
FT_LOSS_DEFAULTS = dict(use_logits_loss=True, use_probs_loss=False)


class FTCLIPClassifier(FSCLIPClassifier):
    """Fine-tuned CLIP for few-shot classification (clip_cls_ft.py:15-333)."""

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
        self._tower = None        # eventclip_amd.ft.VisualTower while an FTTrainer is attached
        self._trainer = None

    def _view_feats(self, data_dict):
        if self._tower is None or 'patches' not in data_dict:
            return super()._view_feats(data_dict)
        feats = self._tower.encode_patches(data_dict['patches'])
        return feats.float().contiguous(), data_dict['row_idx'].contiguous(), data_dict['valid_mask']

    def get_img_feats(self, imgs):
        if self._tower is None:
            return super().get_img_feats(imgs)
        from . import _lib
        t = self._tower
        x = imgs.to(t.dev, torch.float32).contiguous()
        patches = torch.empty((x.shape[0], t.G, t.kpad), dtype=t.cd, device=t.dev)
        _lib.check(_lib.lib().ec_patchify(_lib.ptr(x), x.shape[0], t.cfg['image_size'], t.P, t.kpad, _lib.ptr(patches),
                                          t.code, _lib.stream_ptr()), 'ec_patchify')
        return self._adjust_dtype(t.encode_patches(patches))

    # ---- checkpoints: model.visual.* travels with the classifier (clip_cls_ft.py:313-333) ----
    def state_dict(self, *args, **kwargs):
        w = super(FSCLIPClassifier, self).state_dict(*args, **kwargs)    # the ZS filter drops model.*
        if self._trainer is not None:
            vis = {k: v.detach() for k, v in self._trainer.visual_state_dict().items()}
        else:
            vis = {'model.visual.' + k: v for k, v in self.model.visual.state_dict().items()}
        return {**vis, **w}

    def load_state_dict(self, state_dict, strict=True):
        sd = dict(state_dict)
        vis = {k[len('model.visual.'):]: v for k, v in sd.items() if k.startswith('model.visual.')}
        rest = {k: v for k, v in sd.items() if not k.startswith('model.')}
        if vis and self._trainer is not None:
            # the resume flow (nerv builds the optimiser first, then loads the checkpoint): the weights go INTO the
            # trainer -- masters in place, LoRA factors as factors (not folded into an already-merged base) --
            # and its 16-bit operand copies are rebuilt
            self._trainer.load_visual_state_dict(vis, strict=strict)
        elif vis:
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
        """clip_cls_ft.py:300-306: CLIP stays in eval mode except for its vision tower -- which has no
        train-mode behaviour of its own (no dropout, no batch statistics)."""
        return super().train(mode)
