"""Oracle for fine-tuning the vision tower (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates one optimisation step of the reference's FTCLIPClassifier (/root/reference/models/clip_cls_ft.py)
with torch autograd over the oracle's own functional tower (oracle/clip_ref.py):
  which tensors train     _build_clip, clip_cls_ft.py:44-80 (`lora`, `only_conv1`, `only_bias`, `only_ln`,
                          `only_cls_fc`, `only_cls_token`, else all of model.visual);
  LoRA                    models/lora.py: every attention block's in_proj_weight becomes
                          merged_proj + [up_q down_q; up_k down_k; up_v down_v] (:138-150, k only with
                          'k' in the spec), out_proj.weight becomes linear.weight + up down with 'o' (:50-52);
                          factors start at N(0, 1/r) / zeros (:8-11); state-dict names as the injected
                          modules register them (:384-403);
  forward + loss          clip_cls_ft.py:196-256: encode the valid views, scatter, F.normalize, mask, logits
                          against F.normalize(text_feats) (or fixed text features), aggregate, cross-entropy
                          on the aggregated logits or NLL of log(probs + 1e-6);
  update                  torch.optim.Adam with two learning rates (method.py:152-186).
Pinned by tests/golden/ft_train.npz, produced by the reference's own classes under torch autograd with an
nn.Module vision tower of OpenAI's structure (tools/make_golden_ft_train.py).  The tower arithmetic itself is
oracle/clip_ref.py's (PARITY UNPINNED there: openai/CLIP is un-vendored).
"""
import re

import torch
import torch.nn.functional as F

from . import clip_ref

_IN = re.compile(r'^(transformer\.resblocks\.\d+\.attn)\.in_proj_weight$')
_OUT = re.compile(r'^(transformer\.resblocks\.\d+\.attn)\.out_proj\.(weight|bias)$')


def parse_lora(spec):
    """lora.py:352-365: an int r (> 0) means q, k, v; a string 'qv-16' / 'qkv-16' / 'qkvo-16' names the
    projections.  Returns None (no LoRA) or (r, lora_k, lora_o)."""
    if isinstance(spec, str):
        assert 'q' in spec and 'v' in spec
        return int(spec.split('-')[-1]), 'k' in spec, 'o' in spec
    if spec is None or spec <= 0:
        return None
    return int(spec), True, False


def inject_lora(visual_sd, spec, generator=None):
    """Plain visual state dict (keys without the `visual.` prefix) -> the LoRA-injected one, with the
    reference's key names and initial values (lora.py:8-11: down ~ N(0, 1/r), up = 0)."""
    cfg = parse_lora(spec)
    if cfg is None:
        return dict(visual_sd)
    r, lora_k, lora_o = cfg
    out = {}
    for k, v in visual_sd.items():
        m = _IN.match(k)
        if m:
            W = v.shape[1]
            out[k + '.merged_proj'] = v
            for name in ('q', 'v') + (('k',) if lora_k else ()):
                out[f'{k}.lora_down_{name}'] = torch.randn(r, W, generator=generator, dtype=v.dtype) / r
                out[f'{k}.lora_up_{name}'] = torch.zeros(v.shape[0] // 3, r, dtype=v.dtype)
            continue
        m = _OUT.match(k)
        if m and lora_o:
            out[f'{m.group(1)}.out_proj.linear.{m.group(2)}'] = v
            if m.group(2) == 'weight':
                out[f'{m.group(1)}.out_proj.lora_down.weight'] = torch.randn(r, v.shape[1], generator=generator,
                                                                            dtype=v.dtype) / r
                out[f'{m.group(1)}.out_proj.lora_up.weight'] = torch.zeros(v.shape[0], r, dtype=v.dtype)
            continue
        out[k] = v
    return out


def effective_visual(sd):
    """LoRA-injected (or plain) visual state dict -> OpenAI keys holding the EFFECTIVE weights, built with
    differentiable torch ops on the leaves (lora.py:138-150, :50-52)."""
    out = {}
    for k, v in sd.items():
        if '.lora_' in k:
            continue
        if k.endswith('.in_proj_weight.merged_proj'):
            base = k[:-len('.merged_proj')]
            d = v.shape[0] // 3
            parts = []
            for j, name in enumerate('qkv'):
                w = v[j * d:(j + 1) * d]
                if f'{base}.lora_up_{name}' in sd:
                    w = w + sd[f'{base}.lora_up_{name}'] @ sd[f'{base}.lora_down_{name}']
                parts.append(w)
            out[base] = torch.cat(parts, dim=0)
        elif k.endswith('.out_proj.linear.weight'):
            base = k[:-len('.linear.weight')]
            out[base + '.weight'] = v + sd[base + '.lora_up.weight'] @ sd[base + '.lora_down.weight']
        elif k.endswith('.out_proj.linear.bias'):
            out[k[:-len('.linear.bias')] + '.bias'] = v
        else:
            out[k] = v
    return out


def trainable_names(sd, clip_dict):
    """Which keys of the (LoRA-injected) visual state dict train (clip_cls_ft.py:44-80)."""
    lora = clip_dict.get('lora', -1)
    names = set()
    if parse_lora(lora) is not None:
        names |= {k for k in sd if '.lora_' in k}
    conv1, bias, ln = clip_dict.get('only_conv1', False), clip_dict.get('only_bias', False), clip_dict.get('only_ln', False)
    cls_fc, cls_token = clip_dict.get('only_cls_fc', False), clip_dict.get('only_cls_token', False)
    if conv1:
        names.add('conv1.weight')
    if bias:
        names |= {k for k in sd if 'bias' in k}
    if ln:
        names |= {k for k in sd if re.search(r'(^|\.)ln_(pre|post|1|2)\.(weight|bias)$', k)}
    if cls_fc:
        names.add('proj')
    if cls_token:
        names.add('class_embedding')
    if parse_lora(lora) is None and not (conv1 or bias or ln or cls_fc or cls_token):
        names = set(sd)
    return names


def head(img_feats, valid, labels, text, logit_scale, agg, probs_loss, normalize_text):
    """clip_cls_ft.py:196-256 after the encoder.  img_feats [Nv, D] of the valid views in (b, t) order."""
    B, T = valid.shape
    full = torch.zeros(B, T, img_feats.shape[-1], dtype=img_feats.dtype)
    full[valid] = img_feats
    full = F.normalize(full, p=2, dim=-1) * valid.to(img_feats.dtype).unsqueeze(-1)
    t = F.normalize(text, p=2, dim=-1) if normalize_text else text
    L = logit_scale * full @ t.T
    n = valid.to(img_feats.dtype).sum(1, keepdim=True)
    if agg == 'sum':
        logits = L.sum(1)
    elif agg == 'mean':
        logits = L.sum(1) / n
    else:
        raise NotImplementedError(agg)
    probs = (L.softmax(-1) * valid.to(img_feats.dtype)[..., None]).sum(1) / n
    if probs_loss:
        loss = F.nll_loss((probs + 1e-6).log(), labels)
    else:
        loss = F.cross_entropy(logits, labels)
    return loss, dict(full_logits=L, logits=logits, probs=probs)


def loss_and_grads(visual_sd, cfg, imgs, valid, labels, text, logit_scale, agg='mean', probs_loss=False,
                   train=None, text_trainable=True, dtype=torch.float64):
    """visual_sd: (LoRA-injected or plain) visual state dict; imgs [B, T, 3, R, R]; text [K, D] (the raw
    `text_feats` parameter when text_trainable, else fixed normalised features).  train: names that require
    grad (default: all).  Returns (loss, grads {name: tensor; 'text_feats'}, out dict, image features)."""
    leaves = {k: v.detach().to(dtype).clone().requires_grad_(train is None or k in train) for k, v in visual_sd.items()}
    t = text.detach().to(dtype).clone().requires_grad_(bool(text_trainable))
    eff = {'visual.' + k: v for k, v in effective_visual(leaves).items()}
    feats = clip_ref.encode_image_autograd(eff, cfg, imgs[valid].to(dtype))
    loss, out = head(feats, valid, labels.long(), t, logit_scale, agg, probs_loss, text_trainable)
    loss.backward()
    grads = {k: v.grad for k, v in leaves.items() if v.grad is not None}
    if text_trainable:
        grads['text_feats'] = t.grad
    return float(loss.detach()), grads, {k: v.detach() for k, v in out.items()}, feats.detach()
