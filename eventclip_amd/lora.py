"""Serving LoRA fine-tuned vision towers (models/lora.py, models/clip_cls_ft.py) on the MI355X path.

The reference fine-tunes CLIP's vision tower by replacing every nn.MultiheadAttention with a LoRA-injected
one (``inject_trainable_lora``, lora.py:384-403) and saves `model.visual.*` in the checkpoint
(clip_cls_ft.py:313-321).  At inference the low-rank factors only ever act through the effective weights
``W + up @ down`` (lora.py:138-150 for the merged q / k / v projection, :50-52 for out_proj), so they are
folded into plain OpenAI-layout tensors once on the host and the tower runs unchanged (same kernels, same
speed).  ``merge_lora_visual`` does the fold; ``load_finetuned_visual`` applies it to a CLIP state dict.
"""
import re

import torch

_Q = re.compile(r'^(?P<pre>.*attn)\.in_proj_weight\.merged_proj$')
_O = re.compile(r'^(?P<pre>.*attn)\.out_proj\.linear\.weight$')


def merge_lora_visual(sd):
    """sd: the `visual` part of a fine-tuned checkpoint (keys as under ``model.visual.``, LoRA-injected or
    not).  Returns a state dict with the plain OpenAI keys (`...attn.in_proj_weight`, `...attn.out_proj.weight`,
    `...attn.out_proj.bias`) holding the effective weights; every other tensor is passed through."""
    out = {}
    for k, v in sd.items():
        if '.lora_' in k or k.endswith('.merged_proj') or '.out_proj.linear.' in k:
            continue
        out[k] = v
    for k, v in sd.items():
        m = _Q.match(k)
        if m:
            pre = m.group('pre') + '.in_proj_weight.'
            d = v.shape[0] // 3
            w = v.clone().float()
            for j, name in enumerate('qkv'):                       # lora.py:140-149: k only when lora_k
                if pre + 'lora_up_' + name in sd:
                    w[j * d:(j + 1) * d] += sd[pre + 'lora_up_' + name].float() @ sd[pre + 'lora_down_' + name].float()
            out[m.group('pre') + '.in_proj_weight'] = w
        m = _O.match(k)
        if m:
            pre = m.group('pre') + '.out_proj.'
            out[pre + 'weight'] = v.float() + sd[pre + 'lora_up.weight'].float() @ sd[pre + 'lora_down.weight'].float()
            if pre + 'linear.bias' in sd:
                out[pre + 'bias'] = sd[pre + 'linear.bias']
    return out


def load_finetuned_visual(clip_state_dict, checkpoint):
    """CLIP state dict (OpenAI keys) with its `visual.*` entries replaced by the fine-tuned, LoRA-folded
    ones of ``checkpoint`` (a nerv-style ``{'state_dict': ...}`` or a bare state dict whose vision-tower keys
    start with ``model.visual.``, clip_cls_ft.py:313-321).  Other checkpoint entries (`text_feats`,
    `adapter.*`) belong to the classifier's own ``load_state_dict``."""
    sd = checkpoint.get('state_dict', checkpoint)
    vis = {k[len('model.visual.'):]: v for k, v in sd.items() if k.startswith('model.visual.')}
    if not vis:
        raise KeyError('no model.visual.* entries in the checkpoint')
    merged = merge_lora_visual(vis)
    out = {k: v for k, v in clip_state_dict.items() if not k.startswith('visual.')}
    out.update({'visual.' + k: torch.as_tensor(v) for k, v in merged.items()})
    return out
