"""Golden vectors for serving fine-tuned checkpoints, from the REFERENCE's own FTCLIPClassifier
(models/clip_cls_ft.py with models/lora.py; `clip` and `nerv` stood in as in make_golden_models.py; build
container only).  A small stand-in `clip_model` whose `.visual` holds real nn.MultiheadAttention blocks
goes through the reference class once fully fine-tuned (lora = -1) and once with `lora='qkvo-2'`
(`inject_trainable_lora`); "training" is simulated by perturbing what the class marked trainable.
Stored: the checkpoint exactly as `FTCLIPClassifier.state_dict()` writes it (`model.visual.*`, LoRA keys
included, `text_feats`, `adapter.dummy`), the inputs, the forward outputs, and the output of the
(LoRA-injected) visual stand-in on a probe, which the folded plain weights must reproduce.
Writes tests/golden/classify_ft.npz.

    python tools/make_golden_ft.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, 'tests', 'golden')

clip_stub = types.ModuleType('clip')
clip_stub.tokenize = lambda s: torch.tensor([[abs(hash(s)) % 97 + 1] + [0] * 76])
sys.modules['clip'] = clip_stub
nerv = types.ModuleType('nerv')
nerv_training = types.ModuleType('nerv.training')
nerv_training.BaseModel = nn.Module
sys.modules['nerv'] = nerv
sys.modules['nerv.training'] = nerv_training


def load_ref(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


pkg = types.ModuleType('refmodels')
pkg.__path__ = ['/root/reference/models']
sys.modules['refmodels'] = pkg
load_ref('refmodels.adapter', '/root/reference/models/adapter.py')
load_ref('refmodels.lora', '/root/reference/models/lora.py')
ref_ft = load_ref('refmodels.clip_cls_ft', '/root/reference/models/clip_cls_ft.py')


class Block(nn.Module):
    def __init__(self, d, heads):
        super().__init__()
        self.attn = nn.MultiheadAttention(d, heads)
        self.ln_1 = nn.LayerNorm(d)


class Visual(nn.Module):
    """Enough of CLIP's VisionTransformer for the class: attention blocks to inject LoRA into, a conv,
    LayerNorms, a projection, `output_dim`; `forward` mixes a probe through the attention blocks."""

    def __init__(self, d=16, heads=2, layers=2, out=8):
        super().__init__()
        self.conv1 = nn.Conv2d(3, d, 2, 2, bias=False)
        self.class_embedding = nn.Parameter(torch.randn(d) * 0.1)
        self.transformer = nn.Module()
        self.transformer.resblocks = nn.Sequential(*[Block(d, heads) for _ in range(layers)])
        self.ln_post = nn.LayerNorm(d)
        self.proj = nn.Parameter(torch.randn(d, out) * 0.2)
        self.output_dim = out

    def forward(self, x):                      # x [S, N, d]
        for blk in self.transformer.resblocks:
            h = blk.ln_1(x)
            x = x + blk.attn(h, h, h, need_weights=False)[0]
        return self.ln_post(x[0]) @ self.proj


class FakeCLIP(nn.Module):
    def __init__(self, C, table):
        super().__init__()
        self.logit_scale = nn.Parameter(torch.tensor(float(np.log(100.))))
        self.table = nn.Parameter(table)
        self.visual = Visual(out=C)
        self.C = C

    def encode_image(self, imgs):
        return imgs.flatten(1)[:, :self.C] * 1.5

    def encode_text(self, tokens):
        return self.table[tokens[:, 0].long()]


def main():
    torch.manual_seed(7)
    C, K, R = 8, 6, 4
    table = torch.nn.functional.normalize(torch.randn(100, C), dim=-1) * 3
    names = [f'class_{i}' for i in range(K)]
    tokens = torch.cat([clip_stub.tokenize('a point cloud image of a {}'.format(
        c.lower().replace('_', ' '))) for c in names])
    B, T = 4, 3
    valid = torch.rand(B, T) < 0.6
    valid[:, 0] = True
    imgs = torch.randn(B, T, 3, R, R) * valid[:, :, None, None, None]
    probe = torch.randn(5, 2, 16)
    out = dict(C=C, K=K, imgs=imgs.numpy(), valid=valid.numpy(), table=table.numpy(), tokens=tokens.numpy(),
               probe=probe.numpy())
    base_clip = FakeCLIP(C, table.clone())
    out.update({'base:' + k: v.detach().numpy() for k, v in base_clip.state_dict().items()})
    for tag, lora in (('full', -1), ('lora', 'qkvo-2')):
        for agg in ('sum', 'mean'):
            torch.manual_seed(11)
            clip_model = FakeCLIP(C, table.clone())
            clip_model.load_state_dict(base_clip.state_dict())
            model = ref_ft.FTCLIPClassifier(
                adapter_dict=dict(adapter_type='text-identity', residual=True),
                clip_dict=dict(clip_model=clip_model, prompt='a point cloud image of a {}', class_names=names,
                               agg_func=agg, lora=lora, only_conv1=False, only_bias=False, only_ln=False),
                loss_dict=dict(use_logits_loss=True, use_probs_loss=False))
            trainable = sorted(n for n, p in model.named_parameters() if p.requires_grad)
            with torch.no_grad():
                for n, p in model.named_parameters():        # "training": move what the class unfroze
                    if p.requires_grad:
                        p.add_(torch.randn_like(p) * 0.1)
            model.eval()
            with torch.no_grad():
                o = model({'img': imgs, 'valid_mask': valid})
                vis_out = model.model.visual(probe)
            sd = model.state_dict()
            assert all(not k.startswith('model.') or k.startswith('model.visual.') for k in sd)
            if agg == 'sum':
                out[f'{tag}/trainable'] = np.array(trainable)
                out[f'{tag}/visual_out'] = vis_out.numpy()
                for k, v in sd.items():
                    out[f'{tag}/sd:{k}'] = v.numpy()
            for k in ('full_logits', 'logits', 'probs'):
                out[f'{tag}/{agg}_{k}'] = o[k].numpy()
    path = os.path.join(GOLD, 'classify_ft.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB')
    print('full trainable:', len(out['full/trainable']), 'lora trainable:', list(out['lora/trainable'])[:6], '...')
    print('lora checkpoint keys (block 0):', sorted(k for k in out if k.startswith('lora/sd:model.visual.transformer.resblocks.0')))


if __name__ == '__main__':
    main()
