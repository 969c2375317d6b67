"""Golden vectors for serving LoRA fine-tuned checkpoints (models/lora.py, models/clip_cls_ft.py), from the
REFERENCE's own models/lora.py (torch only; build container): a small stand-in for `model.visual`
(`transformer.resblocks[i].attn = nn.MultiheadAttention`) gets `inject_trainable_lora`, its LoRA factors are
randomised, and the fixture holds its state dict (the keys an FTCLIPClassifier checkpoint carries under
`model.visual.`) together with the effective weights the reference's modules compute (`in_proj_weight()`,
`out_proj.weight`) and an attention output.  Writes tests/golden/lora.npz.

    python tools/make_golden_lora.py
"""
import importlib.util
import os

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location('ref_lora', '/root/reference/models/lora.py')
lora = importlib.util.module_from_spec(spec)
spec.loader.exec_module(lora)


class Block(nn.Module):
    def __init__(self, d, heads):
        super().__init__()
        self.attn = nn.MultiheadAttention(d, heads)


class Visual(nn.Module):
    def __init__(self, d=16, heads=2, layers=2):
        super().__init__()
        self.transformer = nn.Module()
        self.transformer.resblocks = nn.Sequential(*[Block(d, heads) for _ in range(layers)])


def main():
    out = {}
    for tag, r in (('r4', 4), ('qv', 'qv-3'), ('qkvo', 'qkvo-2')):
        torch.manual_seed(3)
        vis = Visual()
        base = {k: v.clone() for k, v in vis.state_dict().items()}
        vis = lora.inject_trainable_lora(vis, r=r)
        with torch.no_grad():
            for k, p in vis.named_parameters():
                if 'lora_' in k:
                    p.copy_(torch.randn_like(p) * 0.1)
        sd = vis.state_dict()
        for k, v in sd.items():
            out[f'{tag}/sd:{k}'] = v.numpy()
        for k, v in base.items():
            out[f'{tag}/base:{k}'] = v.numpy()
        x = torch.randn(5, 2, 16)
        for i, blk in enumerate(vis.transformer.resblocks):
            att = blk.attn
            out[f'{tag}/eff:transformer.resblocks.{i}.attn.in_proj_weight'] = att.in_proj_weight().detach().numpy()
            out[f'{tag}/eff:transformer.resblocks.{i}.attn.out_proj.weight'] = att.out_proj.weight.detach().numpy()
            with torch.no_grad():
                out[f'{tag}/y{i}'] = att.eval()(x, x, x, need_weights=False)[0].numpy()
        out[f'{tag}/x'] = x.numpy()
    path = os.path.join(ROOT, 'tests', 'golden', 'lora.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB')
    print(sorted(k for k in out if k.startswith('qkvo/sd:') and 'resblocks.0' in k))


if __name__ == '__main__':
    main()
