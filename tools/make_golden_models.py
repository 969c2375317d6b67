"""Golden vectors from the REFERENCE's own models/adapter.py and models/clip_cls.py
(imported from /root/reference with in-memory stubs for the un-vendored `clip` and
`nerv` packages; build container only).  Writes tests/golden/adapter_small.npz and
tests/golden/classify_{zs,fs}.npz: inputs, seeded weights and the reference outputs.

    python tools/make_golden_models.py
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

# ---- stubs for un-vendored packages (only what models/ touches at import time) ----
clip_stub = types.ModuleType('clip')
clip_stub.tokenize = lambda s: torch.tensor([[abs(hash(s)) % 97 + 1] + [0] * 76])
sys.modules['clip'] = clip_stub
nerv = types.ModuleType('nerv')
nerv_training = types.ModuleType('nerv.training')
nerv_training.BaseModel = nn.Module
sys.modules['nerv'] = nerv
sys.modules['nerv.training'] = nerv_training


def load_ref(name, path, package=None):
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=None)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


pkg = types.ModuleType('refmodels')
pkg.__path__ = ['/root/reference/models']
sys.modules['refmodels'] = pkg
ref_adapter = load_ref('refmodels.adapter', '/root/reference/models/adapter.py')
ref_cls = load_ref('refmodels.clip_cls', '/root/reference/models/clip_cls.py')


class FakeCLIP(nn.Module):
    """Deterministic stand-in with the protocol clip_cls.py needs."""

    def __init__(self, C, table):
        super().__init__()
        self.logit_scale = nn.Parameter(torch.tensor(float(np.log(100.))))
        self.table = nn.Parameter(table)             # [vocab, C] "text tower"
        self.visual = types.SimpleNamespace(output_dim=C)
        self.C = C

    def encode_image(self, imgs):
        return imgs.flatten(1)[:, :self.C] * 1.5

    def encode_text(self, tokens):
        return self.table[tokens[:, 0].long()]


def adapter_fixture():
    torch.manual_seed(0)
    cfg = dict(in_dim=64, d_model=32, num_heads=2, ffn_dim=64, norm_first=True, num_layers=2,
               residual=0.8)
    ad = ref_adapter.TransformerAdapter(**cfg).eval()
    with torch.no_grad():
        for p in ad.parameters():                    # make biases / LN terms non-trivial
            p.add_(torch.randn_like(p) * 0.05)
    out = {'w:' + k: v.numpy() for k, v in ad.state_dict().items()}
    cases = []
    for T in (1, 2, 5, 10):
        B = 4
        valid = torch.rand(B, T) < 0.6
        valid[:, 0] = True
        feats = torch.randn(B, T, cfg['in_dim']) * valid[..., None]
        with torch.no_grad():
            y = ad(feats, valid)
        cases.append(T)
        out[f'feats_T{T}'], out[f'valid_T{T}'], out[f'out_T{T}'] = feats.numpy(), valid.numpy(), y.numpy()
    out['Ts'] = np.array(cases)
    for k, v in cfg.items():
        out['cfg_' + k] = np.array(v)
    np.savez_compressed(os.path.join(GOLD, 'adapter_small.npz'), **out)

    # full-size default config: only asserted here (too big to store)
    from oracle import adapter as oa
    ad2 = ref_adapter.TransformerAdapter(in_dim=768, residual=0.95).eval()
    valid = torch.rand(3, 10) < 0.5
    valid[:, 0] = True
    feats = torch.randn(3, 10, 768) * valid[..., None]
    with torch.no_grad():
        want = ad2(feats, valid)
    got = oa.transformer_adapter(ad2.state_dict(), feats, valid, 4, 0.95)
    err = ((got - want).abs() * valid[..., None]).max().item()
    print('oracle adapter vs reference (768-dim, valid rows): max abs', err)
    assert err < 1e-5


def classify_fixtures():
    torch.manual_seed(1)
    C, K, R = 32, 7, 4
    table = torch.nn.functional.normalize(torch.randn(100, C), dim=-1) * 3
    names = [f'class_{i}' for i in range(K)]
    tokens = torch.cat([clip_stub.tokenize('a point cloud image of a {}'.format(
        c.lower().replace('_', ' '))) for c in names])
    B, T = 5, 4
    valid = torch.rand(B, T) < 0.6
    valid[:, 0] = True
    imgs = torch.randn(B, T, 3, R, R) * valid[:, :, None, None, None]
    zs = dict(C=C, K=K, imgs=imgs.numpy(), valid=valid.numpy(), table=table.numpy(),
              tokens=tokens.numpy())
    for agg in ('sum', 'mean', 'max'):
        model = ref_cls.ZSCLIPClassifier(clip_dict=dict(
            clip_model=FakeCLIP(C, table.clone()), prompt='a point cloud image of a {}',
            class_names=names, agg_func=agg)).eval()
        try:
            with torch.no_grad():
                o = model({'img': imgs, 'valid_mask': valid})
        except RuntimeError as e:
            # the reference's 'max' aggregation subtracts a [B, T] mask from [B, T, K]
            # logits without unsqueezing (clip_cls.py:117) and cannot broadcast
            assert agg == 'max', e
            zs['max_raises_in_reference'] = np.array(True)
            continue
        for k in ('full_logits', 'logits', 'probs'):
            zs[f'{agg}_{k}'] = o[k].numpy()
    np.savez_compressed(os.path.join(GOLD, 'classify_zs.npz'), **zs)

    fs = dict(C=C, K=K, imgs=imgs.numpy(), valid=valid.numpy(), table=table.numpy(),
              tokens=tokens.numpy())
    ad_cfg = dict(adapter_type='text-trans', in_dim=C, d_model=32, num_heads=2, ffn_dim=64,
                  norm_first=True, num_layers=2, residual=0.8)
    for agg in ('sum', 'mean', 'max'):
        torch.manual_seed(2)
        model = ref_cls.FSCLIPClassifier(
            adapter_dict=dict(ad_cfg),
            clip_dict=dict(clip_model=FakeCLIP(C, table.clone()),
                           prompt='a point cloud image of a {}', class_names=names, agg_func=agg),
            loss_dict=dict(use_logits_loss=True, use_probs_loss=False)).eval()
        with torch.no_grad():
            model.text_feats.add_(torch.randn_like(model.text_feats) * 0.1)   # "tuned" prompts
        if agg == 'max':
            continue                                  # raises in the reference, see above
        with torch.no_grad():
            o = model({'img': imgs, 'valid_mask': valid})
        sd = model.state_dict()
        assert not any(k.startswith('model.') for k in sd) and len(sd) == 29, len(sd)
        if agg == 'sum':
            for k, v in sd.items():
                fs['w:' + k] = v.numpy()
        for k in ('full_logits', 'logits', 'probs'):
            fs[f'{agg}_{k}'] = o[k].numpy()
    for k, v in ad_cfg.items():
        fs['adcfg_' + k] = np.array(v)
    np.savez_compressed(os.path.join(GOLD, 'classify_fs.npz'), **fs)


if __name__ == '__main__':
    adapter_fixture()
    classify_fixtures()
    for f in ('adapter_small.npz', 'classify_zs.npz', 'classify_fs.npz'):
        print(f, os.path.getsize(os.path.join(GOLD, f)) // 1024, 'KiB')
