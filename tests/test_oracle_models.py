"""Adapter / classifier oracles against vectors produced by the reference's own classes."""
import os

import numpy as np
import torch

from conftest import GOLDEN


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def fake_feats(imgs, C):
    # the FakeCLIP.encode_image of tools/make_golden_models.py
    return imgs.flatten(1)[:, :C] * 1.5


def test_adapter_oracle_matches_reference():
    from oracle import adapter as oa
    z = load('adapter_small.npz')
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('w:')}
    for T in z['Ts']:
        feats = torch.from_numpy(z[f'feats_T{T}'])
        valid = torch.from_numpy(z[f'valid_T{T}'])
        got = oa.transformer_adapter(sd, feats, valid, int(z['cfg_num_heads']),
                                     float(z['cfg_residual']))
        want = torch.from_numpy(z[f'out_T{T}'])
        m = valid[..., None].float()       # padded rows are masked downstream (clip_cls.py:329)
        torch.testing.assert_close(got * m, want * m, rtol=1e-5, atol=1e-6)


def test_zero_shot_oracle_matches_reference():
    from oracle import classify as oc
    z = load('classify_zs.npz')
    C = int(z['C'])
    imgs, valid = torch.from_numpy(z['imgs']), torch.from_numpy(z['valid'])
    table, tokens = torch.from_numpy(z['table']), torch.from_numpy(z['tokens'])
    text = torch.nn.functional.normalize(table[tokens[:, 0].long()], dim=-1)
    feats = fake_feats(imgs[valid], C)
    assert bool(z['max_raises_in_reference'])
    for agg in ('sum', 'mean'):
        o = oc.zs_forward(feats, valid, text, 100.0, agg)
        for k in ('full_logits', 'logits', 'probs'):
            torch.testing.assert_close(o[k], torch.from_numpy(z[f'{agg}_{k}']), rtol=1e-5,
                                       atol=1e-4)
    # 'max' (broken upstream): the intended semantics ignore padded views
    o = oc.zs_forward(feats, valid, text, 100.0, 'max')
    full = o['full_logits'].clone()
    full[~valid] = -1e9
    torch.testing.assert_close(o['logits'], full.max(1)[0])


def test_few_shot_tail_oracle_matches_reference():
    from oracle import adapter as oa
    from oracle import classify as oc
    z = load('classify_fs.npz')
    C = int(z['C'])
    imgs, valid = torch.from_numpy(z['imgs']), torch.from_numpy(z['valid'])
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('w:')}
    assert len(sd) == 29 and 'text_feats' in sd     # adapter.* + text_feats, no model.* keys
    B, T = valid.shape
    full = torch.zeros(B, T, C)
    full[valid] = fake_feats(imgs[valid], C)
    ad = oa.transformer_adapter(sd, full, valid, int(z['adcfg_num_heads']),
                                float(z['adcfg_residual']), prefix='adapter.')
    text = torch.nn.functional.normalize(sd['text_feats'], dim=-1)
    for agg in ('sum', 'mean'):
        o = oc.fs_tail(ad, valid, text, 100.0, agg)
        for k in ('full_logits', 'logits', 'probs'):
            torch.testing.assert_close(o[k], torch.from_numpy(z[f'{agg}_{k}']), rtol=1e-4,
                                       atol=1e-4)
