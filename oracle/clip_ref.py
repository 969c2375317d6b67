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

``encode_image(..., emulate='fp16_reference')`` / ``encode_text(...)`` restate the arithmetic the reference
actually runs on its GPU (/root/reference/test.py:25-26: ``clip.load(arch, 'cuda')`` = openai/CLIP's
``convert_weights``: Conv / Linear / MultiheadAttention weights and biases, ``proj`` and ``text_projection`` in
fp16; its ``LayerNorm`` subclass computes in fp32 and casts back; every activation, the residual stream
included, is an fp16 tensor; GEMMs and softmax accumulate in fp32 and round their result once).  It is the
YARDSTICK of the logit-parity tests: the HIP path's distance from the fp32 oracle must not exceed the
reference's own.  Emulated on the CPU in fp32 with an explicit round-to-fp16 after every op, so sums
associate differently from cuBLAS -- a property of any two GPU libraries too.
"""
import torch
import torch.nn.functional as F


def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


def _h(x):
    """round to fp16 and back: the value an fp16 tensor would hold"""
    return x.half().float()


def _quick_gelu_h(x):
    # x * torch.sigmoid(1.702 * x) on an fp16 tensor: three elementwise kernels, each rounds its result
    return _h(x * _h(torch.sigmoid(_h(1.702 * x))))


def _ln_h(x, w, b):
    # openai/CLIP LayerNorm.forward: super().forward(x.type(torch.float32)) -> .type(orig_type)
    return _h(F.layer_norm(x, (x.shape[-1],), w, b, 1e-5))


def _mha_h(x, w_in, b_in, w_out, b_out, heads, mask=None):
    """F.multi_head_attention_forward on fp16 tensors: every op's result rounded to fp16, fp32 accumulation inside
    the GEMMs / bmm / softmax (weights and biases arrive rounded)."""
    N, S, W = x.shape
    hd = W // heads
    qkv = _h(F.linear(x, w_in, b_in))
    q, k, v = qkv.split(W, dim=-1)
    q = _h(q.view(N, S, heads, hd).transpose(1, 2) * (hd ** -0.5))       # q scaled before the product (exact: 1/8)
    k = k.view(N, S, heads, hd).transpose(1, 2)
    v = v.view(N, S, heads, hd).transpose(1, 2)
    att = _h(q @ k.transpose(-1, -2))
    if mask is not None:
        att = _h(att + mask)
    att = _h(att.softmax(dim=-1))
    out = _h(att @ v).transpose(1, 2).reshape(N, S, W)
    return _h(F.linear(out, w_out, b_out))


def _blocks_h(x, sd, prefix, layers, heads, mask=None):
    for i in range(layers):
        p = f'{prefix}.resblocks.{i}.'
        h = _ln_h(x, sd[p + 'ln_1.weight'], sd[p + 'ln_1.bias'])
        x = _h(x + _mha_h(h, sd[p + 'attn.in_proj_weight'], sd[p + 'attn.in_proj_bias'],
                          sd[p + 'attn.out_proj.weight'], sd[p + 'attn.out_proj.bias'], heads, mask))
        h = _ln_h(x, sd[p + 'ln_2.weight'], sd[p + 'ln_2.bias'])
        h = _quick_gelu_h(_h(F.linear(h, sd[p + 'mlp.c_fc.weight'], sd[p + 'mlp.c_fc.bias'])))
        x = _h(x + _h(F.linear(h, sd[p + 'mlp.c_proj.weight'], sd[p + 'mlp.c_proj.bias'])))
    return x


def fp16_reference_weights(sd):
    """openai/CLIP convert_weights: what becomes fp16 (values kept as fp32 tensors holding fp16 values).  LayerNorm
    terms, class / positional / token embeddings and logit_scale stay fp32 parameters (the embeddings are cast
    with .to(x.dtype) where they are added)."""
    keep32 = ('ln_', 'class_embedding', 'positional_embedding', 'token_embedding', 'logit_scale')
    return {k: (v.float() if any(t in k for t in keep32) else _h(v.float())) for k, v in sd.items()}


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
def encode_image(sd, cfg, image, emulate=None):
    """image float32 [N, 3, R, R] -> [N, embed_dim].  emulate='fp16_reference': the reference's GPU arithmetic
    (module docstring); returns fp32 tensors holding the fp16 values."""
    if emulate is None:
        return encode_image_autograd({k: v.float() for k, v in sd.items()}, cfg, image.float())
    assert emulate == 'fp16_reference', emulate
    sd = fp16_reference_weights(sd)
    W, P = cfg['width'], cfg['patch']
    x = _h(F.conv2d(_h(image.float()), sd['visual.conv1.weight'], stride=P))   # image.type(self.dtype)
    x = x.reshape(x.shape[0], W, -1).permute(0, 2, 1)
    cls = _h(sd['visual.class_embedding']).expand(x.shape[0], 1, W)
    x = _h(torch.cat([cls, x], dim=1) + _h(sd['visual.positional_embedding']))
    x = _ln_h(x, sd['visual.ln_pre.weight'], sd['visual.ln_pre.bias'])
    x = _blocks_h(x, sd, 'visual.transformer', cfg['layers'], W // 64)
    x = _ln_h(x[:, 0, :], sd['visual.ln_post.weight'], sd['visual.ln_post.bias'])
    return _h(x @ sd['visual.proj'])


@torch.no_grad()
def encode_text(sd, cfg, tokens, emulate=None):
    """tokens int [K, ctx] -> [K, embed_dim] (not normalised)."""
    TW, ctx = cfg['text_width'], cfg['context_length']
    if emulate is not None:
        assert emulate == 'fp16_reference', emulate
        sd = fp16_reference_weights(sd)
        tokens = tokens.long()
        x = _h(sd['token_embedding.weight'][tokens])                         # .type(self.dtype)
        x = _h(x + _h(sd['positional_embedding']))
        mask = torch.full((ctx, ctx), float('-inf')).triu_(1)
        x = _blocks_h(x, sd, 'transformer', cfg['text_layers'], cfg['text_heads'], mask)
        x = _ln_h(x, sd['ln_final.weight'], sd['ln_final.bias'])
        x = x[torch.arange(x.shape[0]), tokens.argmax(dim=-1)]
        return _h(x @ sd['text_projection'])
    sd = {k: v.float() for k, v in sd.items()}
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
