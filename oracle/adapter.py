"""Oracle for the few-shot feature adapter (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates /root/reference/models/adapter.py in explicit torch fp32 math over the
module's own state-dict keys: TransformerAdapter.forward (:82-105) = in_proj ->
num_layers x nn.TransformerEncoderLayer(norm_first=True, batch_first=True; torch
defaults ReLU, eps 1e-5, eval so no dropout) with src_key_padding_mask = ~valid ->
out_proj -> Adapter.residual_add (:22-25).  Pinned against the reference class itself
by tools/make_golden_models.py -> tests/golden/adapter_small.npz.
"""
import torch
import torch.nn.functional as F


def residual_value(residual):
    """adapter.py:13-20: bool -> 0.5 / 0.0, float kept."""
    if isinstance(residual, bool):
        return 0.5 if residual else 0.
    return float(residual)


@torch.no_grad()
def transformer_adapter(sd, feats, valid_masks, num_heads, residual, prefix=''):
    """sd: state dict with the reference's keys (optionally prefixed, e.g. 'adapter.').
    feats [B, T, C] fp32, valid_masks [B, T] bool -> [B, T, C]."""
    g = lambda k: sd[prefix + k].float()                                  # noqa: E731
    x = F.linear(feats.float(), g('in_proj.weight'), g('in_proj.bias'))  # adapter.py:95
    B, T, dm = x.shape
    hd = dm // num_heads
    n_layers = len({k.split('.')[len(prefix.split('.')) + 1] for k in sd
                    if k.startswith(prefix + 'transformer_encoder.layers.')})
    key_mask = torch.zeros(B, 1, 1, T)
    key_mask.masked_fill_(~valid_masks[:, None, None, :], float('-inf'))  # adapter.py:98-99
    for i in range(n_layers):
        p = f'transformer_encoder.layers.{i}.'
        y = F.layer_norm(x, (dm,), g(p + 'norm1.weight'), g(p + 'norm1.bias'), 1e-5)
        qkv = F.linear(y, g(p + 'self_attn.in_proj_weight'), g(p + 'self_attn.in_proj_bias'))
        q, k, v = qkv.split(dm, dim=-1)
        q = q.view(B, T, num_heads, hd).transpose(1, 2) * hd ** -0.5
        k = k.view(B, T, num_heads, hd).transpose(1, 2)
        v = v.view(B, T, num_heads, hd).transpose(1, 2)
        att = (q @ k.transpose(-1, -2) + key_mask).softmax(-1)
        o = (att @ v).transpose(1, 2).reshape(B, T, dm)
        x = x + F.linear(o, g(p + 'self_attn.out_proj.weight'), g(p + 'self_attn.out_proj.bias'))
        y = F.layer_norm(x, (dm,), g(p + 'norm2.weight'), g(p + 'norm2.bias'), 1e-5)
        y = F.relu(F.linear(y, g(p + 'linear1.weight'), g(p + 'linear1.bias')))
        x = x + F.linear(y, g(p + 'linear2.weight'), g(p + 'linear2.bias'))
    new = F.linear(x, g('out_proj.weight'), g('out_proj.bias'))          # adapter.py:102
    r = residual_value(residual)
    return feats * r + new * (1. - r)                                     # adapter.py:22-25
