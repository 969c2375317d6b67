"""Zero-shot / few-shot classifiers with the reference's API, running on HIP kernels.

Mirror of /root/reference/models/clip_cls.py: ``ZSCLIPClassifier`` (:14-219) and
``FSCLIPClassifier`` (:222-354) take the same ``clip_dict`` / ``adapter_dict`` /
``loss_dict``, expose ``forward(data_dict) -> out_dict`` with the same four keys,
``get_img_feats``, ``get_text_feats``, ``state_dict`` / ``load_state_dict`` that
leave the frozen CLIP weights out, and ``load_weight``.  The arithmetic (image tower,
text tower, adapter, logits, aggregation) is done by libeventclip_hip.so; torch is
used for tensors and the nn.Module plumbing only.  ``forward`` is the inference path; the
training losses of the reference (calc_train_loss, :164-175) and their gradients live in
``eventclip_amd.train`` (cached-feature few-shot training).

``forward`` accepts the reference's batch (``img`` [B, T, 3, R, R] + ``valid_mask``)
or the fused batch of ``Event2ImagePipeline`` (``patches`` + ``row_idx`` +
``valid_mask``), which skips the padded fp32 image tensor.
"""
import copy

import torch
import torch.nn as nn

from . import _lib
from . import clip as eclip
from .adapter import IdentityAdapter, TransformerAdapter

_AGG = {'sum': _lib.EC_AGG_SUM, 'mean': _lib.EC_AGG_MEAN, 'max': _lib.EC_AGG_MAX}


def _l2_normalize(x):
    """F.normalize(x, p=2, dim=-1): x / max(||x||, 1e-12)."""
    return x / x.norm(dim=-1, keepdim=True).clamp_min(1e-12)


# constructor defaults of the reference (clip_cls.py:17-30, :225-243)
ZS_CLIP_DEFAULTS = dict(clip_model=None, prompt='a point cloud image of a {}', class_names=None, agg_func='sum')
ZS_LOSS_DEFAULTS = dict(use_logits_loss=True, use_probs_loss=False)
FS_ADAPTER_DEFAULTS = dict(adapter_type='trans', residual=True)
FS_LOSS_DEFAULTS = dict(use_logits_loss=False, use_probs_loss=True)
_ADAPTERS = {'identity': IdentityAdapter, 'trans': TransformerAdapter}


class ZSCLIPClassifier(nn.Module):
    """CLIP model for **zero-shot** classification (clip_cls.py:14-219)."""

    def __init__(self, clip_dict=None, loss_dict=None):
        super().__init__()
        self.clip_dict = dict(ZS_CLIP_DEFAULTS) if clip_dict is None else clip_dict
        self.loss_dict = dict(ZS_LOSS_DEFAULTS) if loss_dict is None else loss_dict
        self._build_clip()
        self._build_loss()

    def _build_clip(self):
        cd = self.clip_dict
        clip_model = cd['clip_model']
        for w in clip_model.parameters():                                # frozen encoder, clip_cls.py:41-42
            w.requires_grad_(False)
        self.model = clip_model.eval()
        self.logit_scale = float(clip_model.logit_scale.exp())           # clip_cls.py:44
        self.prompt, self.class_names, self.agg_func = cd['prompt'], cd['class_names'], cd['agg_func']
        if self.agg_func not in _AGG:                                    # clip_cls.py:53
            raise AssertionError(f'agg_func {self.agg_func!r} not in {sorted(_AGG)}')
        # token ids [K, 77] for the class prompts, for when the BPE vocabulary file is not
        # available to clip.tokenize (it does not ship with this repo)
        self.class_tokens = cd.get('class_tokens', None)
        self.text_feats, self._text_t = None, None

    def _build_loss(self):
        ld = self.loss_dict
        self.use_logits_loss, self.use_probs_loss = ld['use_logits_loss'], ld['use_probs_loss']
        if bool(self.use_logits_loss) == bool(self.use_probs_loss):      # exactly one of them, clip_cls.py:58-61
            raise AssertionError('set exactly one of use_logits_loss / use_probs_loss')

    def _same_class_names(self, class_names):
        """Pairwise comparison over the shorter of the two lists, as clip_cls.py:62."""
        return all(a == b for a, b in zip(class_names, self.class_names))

    def _tokenize(self, class_names):
        if self.class_tokens is not None and (class_names is self.class_names or
                                              self._same_class_names(class_names)):
            return self.class_tokens
        names = [c.lower().replace('_', ' ') for c in class_names]      # clip_cls.py:80
        return torch.cat([eclip.tokenize(self.prompt.format(c)) for c in names])

    def get_text_feats(self, class_names=None):
        """Text prompt features, L2-normalised and cached (clip_cls.py:64-93).  Unlike the
        reference this does not crash when called with explicit ``class_names`` and an
        empty cache (its ``no_cls_flag`` is unbound there, :74/:90)."""
        no_cls_flag = class_names is None
        if no_cls_flag:
            class_names = self.class_names
        if (no_cls_flag or self._same_class_names(class_names)) and self.text_feats is not None:
            return self.text_feats
        prompts = self._tokenize(class_names).to(self.device)
        text_feats = self.model.encode_text(prompts)                     # clip_cls.py:84
        text_feats = _l2_normalize(text_feats)                           # clip_cls.py:85
        if no_cls_flag or self._same_class_names(class_names):
            self.text_feats = text_feats
            self._text_t = None
        return text_feats

    def _text_transposed(self):
        t = self.get_text_feats()
        if self._text_t is None or self._text_t.shape[1] != t.shape[0]:
            self._text_t = t.detach().float().t().contiguous()
        return self._text_t

    def get_img_feats(self, imgs):
        """imgs [N, 3, R, R] -> [N, C] (clip_cls.py:95-102)."""
        return self.model.encode_image(imgs)

    # ---- shared pieces of forward ----
    def _view_feats(self, data_dict):
        """Features of the valid views [Nv, C] fp32 plus row_idx [B, T] int32 (CUDA)."""
        valid_masks = data_dict['valid_mask']
        if 'patches' in data_dict:
            feats = self.model.encode_patches(data_dict['patches'])
            row_idx = data_dict['row_idx']
        else:
            imgs = data_dict['img']                                      # [B, T, C, H, W]
            valid_imgs = imgs[valid_masks]                               # clip_cls.py:139
            feats = self.get_img_feats(valid_imgs)
            flat = valid_masks.reshape(-1)
            row_idx = torch.where(flat, torch.cumsum(flat.int(), 0) - 1,
                                  torch.full_like(flat, -1, dtype=torch.int64))
            row_idx = row_idx.to(torch.int32).reshape(valid_masks.shape)
        return feats.float().contiguous(), row_idx.contiguous(), valid_masks

    def _classify(self, feats, row_idx, normalize):
        from . import torch_ops  # noqa: F401  (registers eventclip_hip::classify)
        return torch.ops.eventclip_hip.classify(feats, row_idx, self._text_transposed(),
                                                float(self.logit_scale), _AGG[self.agg_func],
                                                bool(normalize))

    @torch.no_grad()
    def forward(self, data_dict):
        """clip_cls.py:131-162."""
        feats, row_idx, valid_masks = self._view_feats(data_dict)
        # logits = logit_scale * img_feats @ text_feats.T with UN-normalised image feats (:148)
        full_logits, logits, probs = self._classify(feats, row_idx, normalize=False)
        return dict(full_logits=full_logits, valid_masks=valid_masks, logits=logits, probs=probs)

    @torch.no_grad()
    def calc_eval_loss(self, data_dict, out_dict):
        """Accuracies of clip_cls.py:177-192 (the CE terms belong to training)."""
        y = data_dict['label']
        hit = lambda scores: (scores.argmax(-1) == y).float().mean()     # noqa: E731
        return dict(probs_acc=hit(out_dict['probs']), logits_acc=hit(out_dict['logits']))

    dtype = property(lambda self: self.model.logit_scale.dtype)          # clip_cls.py:194-200
    device = property(lambda self: self.model.logit_scale.device)

    def train(self, mode=True):
        super().train(mode)
        self.model.eval()                 # the encoder never leaves eval mode (clip_cls.py:202-206)
        return self

    def state_dict(self, *args, **kwargs):
        """Frozen CLIP weights are not part of a checkpoint (clip_cls.py:208-212)."""
        w = super().state_dict(*args, **kwargs)
        return {k: v for k, v in w.items() if not k.startswith('model.')}

    def load_state_dict(self, state_dict, strict=True):
        merged = {'model.' + k: v for k, v in self.model.state_dict().items()}    # :214-219
        merged.update(state_dict)
        out = super().load_state_dict(merged, strict=strict)
        self._text_t = None
        return out

    def load_weight(self, ckp_path, strict=True):
        """nerv BaseModel.load_weight (test.py:50): checkpoint file with a ``state_dict`` entry."""
        ckp = torch.load(ckp_path, map_location='cpu')
        sd = ckp['state_dict'] if isinstance(ckp, dict) and 'state_dict' in ckp else ckp
        self.load_state_dict(sd, strict=strict)


class FSCLIPClassifier(ZSCLIPClassifier):
    """CLIP model for **few-shot** classification (clip_cls.py:222-354)."""

    def __init__(self, adapter_dict=None, clip_dict=None, loss_dict=None):
        super().__init__(clip_dict=clip_dict, loss_dict=dict(FS_LOSS_DEFAULTS) if loss_dict is None else loss_dict)
        self.adapter_dict = copy.deepcopy(FS_ADAPTER_DEFAULTS if adapter_dict is None else adapter_dict)
        self._build_adapter()

    def _build_prompts(self, adapter_type):
        """'text-xxx': the class prompts' features become a trainable parameter (clip_cls.py:253-259)."""
        with torch.no_grad():
            init = ZSCLIPClassifier.get_text_feats(self).float().clone()  # [n_classes, C]
        self.text_feats = nn.Parameter(init, requires_grad=True)
        return adapter_type[len('text-'):]

    def _build_adapter(self):
        kind = self.adapter_dict.pop('adapter_type').lower()
        self.prompt_tuning = kind.startswith('text-')                    # clip_cls.py:263-266
        if self.prompt_tuning:
            kind = self._build_prompts(kind)
        if kind not in _ADAPTERS:
            raise NotImplementedError(f'adapter {kind} not supported!')
        self.adapter_type = kind
        self.adapter = _ADAPTERS[kind](**self.adapter_dict)

    def _adjust_dtype(self, x):
        """fp16 encoder outputs are widened to the adapter's dtype outside training (clip_cls.py:281-288)."""
        return x if self.training else x.to(self.dtype)

    def get_text_feats(self, class_names=None):
        if self.prompt_tuning:                                           # clip_cls.py:292-295
            return self._adjust_dtype(_l2_normalize(self.text_feats))
        with torch.no_grad():
            return self._adjust_dtype(super().get_text_feats(class_names))

    def _text_transposed(self):
        # learned text features can change under load_state_dict: rebuild when asked
        if self.prompt_tuning:
            return self.get_text_feats().detach().float().t().contiguous()
        return super()._text_transposed()

    @torch.no_grad()
    def get_img_feats(self, imgs):
        return self._adjust_dtype(super().get_img_feats(imgs))

    @torch.no_grad()
    def forward(self, data_dict):
        """clip_cls.py:308-350."""
        feats, row_idx, valid_masks = self._view_feats(data_dict)
        B, T = valid_masks.shape
        C = feats.shape[-1]
        # scatter to [B, T, C] with zero rows for padded views (:319-321), adapter (:322)
        full_img_feats = self.adapter.forward_rows(feats, row_idx)       # [B, T, C] fp32
        # F.normalize + mask + logits + aggregation (:326-343) in ec_classify
        idx = torch.where(valid_masks, torch.arange(B * T, device=feats.device).view(B, T),
                          torch.full((B, T), -1, device=feats.device)).to(torch.int32)
        full_logits, logits, probs = self._classify(full_img_feats.reshape(B * T, C).contiguous(),
                                                    idx.contiguous(), normalize=True)
        return dict(full_logits=full_logits, valid_masks=valid_masks, logits=logits, probs=probs)

    @torch.no_grad()
    def cache_feats(self, data_dict):
        """Frozen-encoder outputs of a batch, scattered to [B, T, C] fp32 with zero rows for padded
        views (clip_cls.py:313-321): what eventclip_amd.train consumes, computed once per sample."""
        feats, row_idx, valid_masks = self._view_feats(data_dict)
        B, T = valid_masks.shape
        full = IdentityAdapter.forward_rows(None, feats.float(), row_idx)
        return full.reshape(B, T, feats.shape[-1]), valid_masks

    dtype = property(lambda self: self.adapter.dtype)


def build_model(params):
    """models/__init__.py:5-21.  'FTCLIP' gives the serving form of FTCLIPClassifier (forward and
    checkpoints; fine-tuning of CLIP itself is out of scope)."""
    kind = params.model
    if kind == 'ZSCLIP':
        return ZSCLIPClassifier(clip_dict=params.clip_dict)
    if kind == 'FSCLIP':
        return FSCLIPClassifier(params.adapter_dict, params.clip_dict, params.loss_dict)
    if kind == 'FTCLIP':
        from .clip_cls_ft import FTCLIPClassifier
        return FTCLIPClassifier(params.adapter_dict, params.clip_dict, params.loss_dict)
    raise NotImplementedError(f'{kind} is not implemented.')
