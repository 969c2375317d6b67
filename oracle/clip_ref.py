"""Oracle for the CLIP towers (TEST INFRASTRUCTURE, see oracle/__init__.py).

PARITY UNPINNED against the reference: the arithmetic of ``encode_image`` /
``encode_text`` (called at /root/reference/models/clip_cls.py:101 and :84) lives in
un-vendored openai/CLIP (``clip==1.0`` from git HEAD, environment.yml:87), whose source
and weights are absent, and the reference has no tests.  This file restates the
published architecture (clip/model.py: VisionTransformer, Transformer,
ResidualAttentionBlock with nn.MultiheadAttention, fp32-computing LayerNorm,
QuickGELU, CLIP.encode_image / encode_text) in plain torch fp32 over a state dict
with OpenAI's key names, and is cross-checked against HF ``transformers`` CLIP with
seeded random weights (tools/make_golden_clip.py; the fixture
tests/golden/clip_tiny.npz carries HF's outputs).
"""
import torch
import torch.nn.functional as F


def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


def _mha(x, w_in, b_in, w_out, b_out, heads, mask=None):
    """nn.MultiheadAttention self-attention on [N, S, W] (batch first), no dropout."""
    N, S, W = x.shape
    hd = W // heads
    qkv = F.linear(x, w_in, b_in)
    q, k, v = qkv.split(W, dim=-1)
    q = q.view(N, S, heads, hd).transpose(1, 2) * (hd ** -0.5)
    k = k.view(N, S, heads, hd).transpose(1, 2)
    v = v.view(N, S, heads, hd).transpose(1, 2)
    att = q @ k.transpose(-1, -2)
    if mask is not None:
        att = att + mask
    att = att.softmax(dim=-1)
    out = (att @ v).transpose(1, 2).reshape(N, S, W)
    return F.linear(out, w_out, b_out)


def _blocks(x, sd, prefix, layers, heads, mask=None):
    W = x.shape[-1]
    for i in range(layers):
        p = f'{prefix}.resblocks.{i}.'
        h = F.layer_norm(x, (W,), sd[p + 'ln_1.weight'], sd[p + 'ln_1.bias'], 1e-5)
        x = x + _mha(h, sd[p + 'attn.in_proj_weight'], sd[p + 'attn.in_proj_bias'],
                     sd[p + 'attn.out_proj.weight'], sd[p + 'attn.out_proj.bias'], heads, mask)
        h = F.layer_norm(x, (W,), sd[p + 'ln_2.weight'], sd[p + 'ln_2.bias'], 1e-5)
        h = quick_gelu(F.linear(h, sd[p + 'mlp.c_fc.weight'], sd[p + 'mlp.c_fc.bias']))
        x = x + F.linear(h, sd[p + 'mlp.c_proj.weight'], sd[p + 'mlp.c_proj.bias'])
    return x


def encode_image_autograd(sd, cfg, image):
    """encode_image without torch.no_grad and in the dtype of ``sd`` (fp32 / fp64 leaves that may require
    grad): what the reference's fine-tuning differentiates (models/clip_cls_ft.py:180-183)."""
    W, P = cfg['width'], cfg['patch']
    image = image.to(sd['visual.conv1.weight'].dtype)
    x = F.conv2d(image, sd['visual.conv1.weight'], stride=P)              # [N, W, g, g]
    x = x.reshape(x.shape[0], W, -1).permute(0, 2, 1)                      # [N, g*g, W]
    cls = sd['visual.class_embedding'].expand(x.shape[0], 1, W)
    x = torch.cat([cls, x], dim=1) + sd['visual.positional_embedding']
    x = F.layer_norm(x, (W,), sd['visual.ln_pre.weight'], sd['visual.ln_pre.bias'], 1e-5)
    x = _blocks(x, sd, 'visual.transformer', cfg['layers'], W // 64)
    x = F.layer_norm(x[:, 0, :], (W,), sd['visual.ln_post.weight'], sd['visual.ln_post.bias'], 1e-5)
    return x @ sd['visual.proj']


@torch.no_grad()
def encode_image(sd, cfg, image):
    """image float32 [N, 3, R, R] -> [N, embed_dim]."""
    return encode_image_autograd({k: v.float() for k, v in sd.items()}, cfg, image.float())


@torch.no_grad()
def encode_text(sd, cfg, tokens):
    """tokens int [K, ctx] -> [K, embed_dim] (not normalised)."""
    sd = {k: v.float() for k, v in sd.items()}
    TW, ctx = cfg['text_width'], cfg['context_length']
    tokens = tokens.long()
    x = sd['token_embedding.weight'][tokens] + sd['positional_embedding']
    mask = torch.full((ctx, ctx), float('-inf')).triu_(1)
    x = _blocks(x, sd, 'transformer', cfg['text_layers'], cfg['text_heads'], mask)
    x = F.layer_norm(x, (TW,), sd['ln_final.weight'], sd['ln_final.bias'], 1e-5)
    x = x[torch.arange(x.shape[0]), tokens.argmax(dim=-1)]
    return x @ sd['text_projection']


def round_weights(sd, dtype):
    """The 16-bit weight rounding the HIP path applies to GEMM operands (LayerNorm terms,
    biases, embeddings and positional tables stay fp32 there)."""
    keep32 = ('ln_', 'bias', 'class_embedding', 'positional_embedding', 'token_embedding',
              'logit_scale')
    out = {}
    for k, v in sd.items():
        if any(t in k for t in keep32):
            out[k] = v.float()
        else:
            out[k] = v.float().to(dtype).float()
    return out
