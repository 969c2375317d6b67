"""Cross-check oracle/clip_ref.py against HF transformers' CLIP (build container only)
and write tests/golden/clip_tiny.npz: a tiny random-weight CLIP in OpenAI's key
layout (fp16-representable values), inputs, and HF's outputs for them.

    python tools/make_golden_clip.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from eventclip_amd import clip as eclip  # noqa: E402
from oracle import clip_ref  # noqa: E402


def hf_model(cfg, sd):
    from transformers import CLIPConfig, CLIPModel
    conf = CLIPConfig(
        vision_config=dict(hidden_size=cfg['width'], intermediate_size=4 * cfg['width'],
                           num_hidden_layers=cfg['layers'], num_attention_heads=cfg['width'] // 64,
                           image_size=cfg['image_size'], patch_size=cfg['patch'],
                           hidden_act='quick_gelu', layer_norm_eps=1e-5,
                           projection_dim=cfg['embed_dim']),
        text_config=dict(hidden_size=cfg['text_width'], intermediate_size=4 * cfg['text_width'],
                         num_hidden_layers=cfg['text_layers'],
                         num_attention_heads=cfg['text_heads'],
                         max_position_embeddings=cfg['context_length'],
                         vocab_size=cfg['vocab_size'], hidden_act='quick_gelu',
                         layer_norm_eps=1e-5, projection_dim=cfg['embed_dim'], eos_token_id=2,
                         bos_token_id=0, pad_token_id=1),
        projection_dim=cfg['embed_dim'])
    m = CLIPModel(conf).eval()
    hf = {}

    def blocks(src, dst, layers, W):
        for i in range(layers):
            s, d = f'{src}.resblocks.{i}.', f'{dst}.encoder.layers.{i}.'
            wq, wk, wv = sd[s + 'attn.in_proj_weight'].split(W, 0)
            bq, bk, bv = sd[s + 'attn.in_proj_bias'].split(W, 0)
            for n, w, b in (('q', wq, bq), ('k', wk, bk), ('v', wv, bv)):
                hf[d + f'self_attn.{n}_proj.weight'], hf[d + f'self_attn.{n}_proj.bias'] = w, b
            hf[d + 'self_attn.out_proj.weight'] = sd[s + 'attn.out_proj.weight']
            hf[d + 'self_attn.out_proj.bias'] = sd[s + 'attn.out_proj.bias']
            for a, b in (('ln_1', 'layer_norm1'), ('ln_2', 'layer_norm2'), ('mlp.c_fc', 'mlp.fc1'),
                         ('mlp.c_proj', 'mlp.fc2')):
                hf[d + b + '.weight'], hf[d + b + '.bias'] = sd[s + a + '.weight'], sd[s + a + '.bias']

    hf['vision_model.embeddings.patch_embedding.weight'] = sd['visual.conv1.weight']
    hf['vision_model.embeddings.class_embedding'] = sd['visual.class_embedding']
    hf['vision_model.embeddings.position_embedding.weight'] = sd['visual.positional_embedding']
    hf['vision_model.pre_layrnorm.weight'] = sd['visual.ln_pre.weight']
    hf['vision_model.pre_layrnorm.bias'] = sd['visual.ln_pre.bias']
    blocks('visual.transformer', 'vision_model', cfg['layers'], cfg['width'])
    hf['vision_model.post_layernorm.weight'] = sd['visual.ln_post.weight']
    hf['vision_model.post_layernorm.bias'] = sd['visual.ln_post.bias']
    hf['visual_projection.weight'] = sd['visual.proj'].t()
    hf['text_model.embeddings.token_embedding.weight'] = sd['token_embedding.weight']
    hf['text_model.embeddings.position_embedding.weight'] = sd['positional_embedding']
    blocks('transformer', 'text_model', cfg['text_layers'], cfg['text_width'])
    hf['text_model.final_layer_norm.weight'] = sd['ln_final.weight']
    hf['text_model.final_layer_norm.bias'] = sd['ln_final.bias']
    hf['text_projection.weight'] = sd['text_projection'].t()
    hf['logit_scale'] = sd['logit_scale']
    missing, unexpected = m.load_state_dict({k: v.clone() for k, v in hf.items()}, strict=False)
    missing = [k for k in missing if 'position_ids' not in k]
    assert not missing and not unexpected, (missing, unexpected)
    return m


def feats(out):
    return out if isinstance(out, torch.Tensor) else out.pooler_output


def main():
    torch.manual_seed(0)
    cfg = dict(image_size=32, patch=8, width=64, layers=2, embed_dim=64, text_width=64,
               text_heads=1, text_layers=2, context_length=77, vocab_size=256)
    sd = eclip.random_state_dict(cfg, seed=7)
    sd = {k: v.half().float() for k, v in sd.items()}          # exactly representable in fp16
    sd['logit_scale'] = torch.tensor(float(np.log(100.0)))
    img = torch.randn(3, 3, 32, 32).half().float()
    tok = torch.zeros(4, 77, dtype=torch.int64)
    g = torch.Generator().manual_seed(1)
    for i in range(4):
        n = 3 + i
        tok[i, 0] = 254
        tok[i, 1:1 + n] = torch.randint(3, 250, (n,), generator=g)
        tok[i, 1 + n] = 255                                     # largest id = EOT
    m = hf_model(cfg, sd)
    with torch.no_grad():
        hf_img = feats(m.get_image_features(pixel_values=img))
        hf_txt = feats(m.get_text_features(input_ids=tok))
    o_img = clip_ref.encode_image(sd, cfg, img)
    o_txt = clip_ref.encode_text(sd, cfg, tok)
    di = (o_img - hf_img).abs().max().item() / hf_img.abs().max().item()
    dt = (o_txt - hf_txt).abs().max().item() / hf_txt.abs().max().item()
    print(f'oracle vs HF: image rel {di:.2e}, text rel {dt:.2e}')
    assert di < 1e-5 and dt < 1e-5
    out = {f'w:{k}': v.half().numpy() for k, v in sd.items() if k != 'logit_scale'}
    out.update(cfg={k: np.array(v) for k, v in cfg.items()})
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'clip_tiny.npz'),
                        cfg_keys=np.array(list(cfg)), cfg_vals=np.array(list(cfg.values())),
                        img=img.half().numpy(), tok=tok.numpy().astype(np.int32),
                        hf_img=hf_img.numpy(), hf_txt=hf_txt.numpy(),
                        **{k: v for k, v in out.items() if k != 'cfg'})
    print('wrote tests/golden/clip_tiny.npz',
          os.path.getsize(os.path.join(ROOT, 'tests', 'golden', 'clip_tiny.npz')) // 1024, 'KiB')

    # full-size single check (not stored): ViT-B/32 block structure against HF
    cfg2 = eclip.arch_config('ViT-B/32', layers=2, text_layers=1, vocab_size=1024)
    sd2 = eclip.random_state_dict(cfg2, seed=3)
    m2 = hf_model(cfg2, sd2)
    img2 = torch.randn(1, 3, 224, 224)
    with torch.no_grad():
        h2 = feats(m2.get_image_features(pixel_values=img2))
    o2 = clip_ref.encode_image(sd2, cfg2, img2)
    print('ViT-B/32 (2 layers) oracle vs HF rel',
          ((o2 - h2).abs().max() / h2.abs().max()).item())


if __name__ == '__main__':
    main()
