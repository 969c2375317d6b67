"""Golden vectors for FINE-TUNING the vision tower, from the REFERENCE's own FTCLIPClassifier
(models/clip_cls_ft.py + models/lora.py under torch autograd; `clip` and `nerv` stood in as in
make_golden_models.py; build container only).

The stand-in `clip_model.visual` is an nn.Module with the published structure of openai/CLIP's
VisionTransformer (conv1, class / positional embeddings, ln_pre, ResidualAttentionBlocks around
nn.MultiheadAttention in [S, N, W] layout, QuickGELU MLP, ln_post, proj), small enough for a fixture but with
the head dim (64) and multiples the HIP kernels need, so the same vectors also drive the GPU tests.  For every
case the reference class decides what trains (`lora`, `only_*`, all), runs forward + calc_train_loss +
backward, and torch.optim.Adam with the two learning rates of method.py:152-186 takes two steps.
Stored: initial state dict (the class's own key names, LoRA keys included; tensors equal to the shared base once), inputs, loss, every gradient,
image features, the out dict, and the parameters after the two steps.  Writes tests/golden/ft_train.npz.

    python tools/make_golden_ft_train.py
"""
import importlib.util
import os
import sys
import types
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, 'tests', 'golden')

clip_stub = types.ModuleType('clip')
clip_stub.tokenize = lambda s: torch.tensor([[sum(s.encode()) % 97 + 1] + [0] * 76])
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

CFG = dict(image_size=8, patch=4, width=64, layers=2, heads=1, embed_dim=16)


class QuickGELU(nn.Module):
    def forward(self, x):
        return x * torch.sigmoid(1.702 * x)


class ResidualAttentionBlock(nn.Module):
    def __init__(self, d, heads):
        super().__init__()
        self.attn = nn.MultiheadAttention(d, heads)
        self.ln_1 = nn.LayerNorm(d)
        self.mlp = nn.Sequential(OrderedDict([('c_fc', nn.Linear(d, d * 4)), ('gelu', QuickGELU()),
                                              ('c_proj', nn.Linear(d * 4, d))]))
        self.ln_2 = nn.LayerNorm(d)

    def forward(self, x):
        h = self.ln_1(x)
        x = x + self.attn(h, h, h, need_weights=False)[0]
        return x + self.mlp(self.ln_2(x))


class Transformer(nn.Module):
    def __init__(self, width, layers, heads):
        super().__init__()
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads) for _ in range(layers)])

    def forward(self, x):
        return self.resblocks(x)


class VisionTransformer(nn.Module):
    def __init__(self, image_size, patch, width, layers, heads, embed_dim):
        super().__init__()
        self.output_dim = embed_dim
        self.conv1 = nn.Conv2d(3, width, patch, patch, bias=False)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn((image_size // patch) ** 2 + 1, width))
        self.ln_pre = nn.LayerNorm(width)
        self.transformer = Transformer(width, layers, heads)
        self.ln_post = nn.LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, embed_dim))

    def forward(self, x):
        x = self.conv1(x)
        x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)
        cls = self.class_embedding.to(x.dtype) + torch.zeros(x.shape[0], 1, x.shape[-1], dtype=x.dtype)
        x = torch.cat([cls, x], dim=1) + self.positional_embedding.to(x.dtype)
        x = self.ln_pre(x)
        x = self.transformer(x.permute(1, 0, 2)).permute(1, 0, 2)
        x = self.ln_post(x[:, 0, :])
        return x @ self.proj


class FakeCLIP(nn.Module):
    def __init__(self, table):
        super().__init__()
        self.logit_scale = nn.Parameter(torch.tensor(float(np.log(100.))))
        self.table = nn.Parameter(table)
        self.visual = VisionTransformer(**CFG)

    def encode_image(self, imgs):
        return self.visual(imgs)

    def encode_text(self, tokens):
        return self.table[tokens[:, 0].long()]


CASES = (
    # tag, clip_dict extras, adapter_type, agg, probs_loss
    ('full', dict(lora=-1), 'text-identity', 'mean', False),
    ('lora_qkvo', dict(lora='qkvo-2'), 'text-identity', 'mean', False),
    ('lora_int', dict(lora=2), 'text-identity', 'sum', False),
    ('lora_qv', dict(lora='qv-3'), 'identity', 'mean', True),
    ('bias', dict(lora=-1, only_bias=True), 'text-identity', 'mean', False),
    ('ln', dict(lora=-1, only_ln=True), 'identity', 'sum', False),
    ('conv_cls', dict(lora=-1, only_conv1=True, only_cls_token=True, only_cls_fc=True), 'text-identity', 'mean', True),
)


def main():
    torch.manual_seed(3)
    D, K, R = CFG['embed_dim'], 5, CFG['image_size']
    table = torch.nn.functional.normalize(torch.randn(100, D), dim=-1)
    names = [f'class_{i}' for i in range(K)]
    B, T = 3, 2
    valid = torch.tensor([[True, True], [True, False], [True, True]])
    imgs = torch.randn(B, T, 3, R, R) * valid[:, :, None, None, None]
    labels = torch.tensor([1, 4, 0])
    base = FakeCLIP(table.clone())
    with torch.no_grad():                      # LayerNorm terms and biases off their trivial initial values
        for n, p in base.visual.named_parameters():
            if 'ln_' in n or 'bias' in n:
                p.add_(torch.randn_like(p) * 0.1)
    out = dict(cfg=np.array([CFG[k] for k in ('image_size', 'patch', 'width', 'layers', 'heads', 'embed_dim')]),
               K=K, imgs=imgs.numpy(), valid=valid.numpy(), labels=labels.numpy(), cases=np.array([c[0] for c in CASES]))
    for k, v in base.visual.state_dict().items():
        out[f'base/sd:model.visual.{k}'] = v.detach().numpy().copy()
    lr, clip_lr = 1e-2, 2e-3
    out['lr'], out['clip_lr'] = lr, clip_lr
    for tag, extra, adapter_type, agg, probs_loss in CASES:
        torch.manual_seed(11)
        clip_model = FakeCLIP(table.clone())
        clip_model.load_state_dict(base.state_dict())
        cd = dict(clip_model=clip_model, prompt='a point cloud image of a {}', class_names=names, agg_func=agg,
                  only_conv1=False, only_bias=False, only_ln=False)
        cd.update(extra)
        model = ref_ft.FTCLIPClassifier(adapter_dict=dict(adapter_type=adapter_type, residual=True), clip_dict=cd,
                                        loss_dict=dict(use_logits_loss=not probs_loss, use_probs_loss=probs_loss))
        with torch.no_grad():                  # LoRA's `up` starts at zero: move it so its partner's gradient is not zero
            for n, p in model.named_parameters():
                if 'lora_up' in n:
                    p.add_(torch.randn_like(p) * 0.05)
        model.train()
        sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
        trainable = sorted(n for n, p in model.named_parameters() if p.requires_grad)
        # method.py:152-186: two groups, everything outside model.visual at lr, model.visual at clip_lr
        adapter_params = [p for n, p in model.named_parameters() if 'model.visual' not in n and p.requires_grad]
        clip_params = [p for n, p in model.named_parameters() if 'model.visual' in n and p.requires_grad]
        opt = torch.optim.Adam([{'params': adapter_params, 'lr': lr}, {'params': clip_params, 'lr': clip_lr}])
        data = {'img': imgs, 'valid_mask': valid, 'label': labels}
        for step in range(2):
            opt.zero_grad()
            o = model(data)
            loss = model.calc_train_loss(data, o)['ce_loss']
            loss.backward()
            if step == 0:
                out[f'{tag}/loss'] = loss.item()
                out[f'{tag}/feats'] = model.get_img_feats(imgs[valid]).detach().numpy()
                for k in ('full_logits', 'logits', 'probs'):
                    out[f'{tag}/{k}'] = o[k].detach().numpy()
                for n, p in model.named_parameters():
                    if p.requires_grad:
                        out[f'{tag}/grad:{n}'] = p.grad.detach().numpy().copy()
            opt.step()
        for n, p in model.named_parameters():      # the parameters after the two steps
            if p.requires_grad:
                out[f'{tag}/step2:{n}'] = p.detach().numpy().copy()
        if not model.prompt_tuning:                # fixed text features: not in the state dict
            out[f'{tag}/text_fixed'] = model.get_text_feats().detach().numpy()
        out[f'{tag}/trainable'] = np.array(trainable)
        out[f'{tag}/agg'], out[f'{tag}/probs_loss'], out[f'{tag}/adapter_type'] = agg, probs_loss, adapter_type
        out[f'{tag}/lora'] = str(extra.get('lora', -1))
        out[f'{tag}/flags'] = np.array([bool(extra.get(k, False)) for k in
                                        ('only_conv1', 'only_bias', 'only_ln', 'only_cls_fc', 'only_cls_token')])
        for k, v in sd0.items():                   # only what differs from the shared base state dict
            if f'base/sd:{k}' not in out or not np.array_equal(out[f'base/sd:{k}'], v.numpy()):
                out[f'{tag}/sd:{k}'] = v.numpy()
        out[f'{tag}/sd_keys'] = np.array(list(sd0.keys()))
        print(tag, 'loss', out[f'{tag}/loss'], 'trainable', len(trainable), trainable[:3])
    path = os.path.join(GOLD, 'ft_train.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()
