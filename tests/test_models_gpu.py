"""Classifier classes and the adapter kernel on the MI355X against the reference's vectors
and the oracles."""
import os
import types

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def fake_clip(C, table):
    """Same stand-in encoder the fixtures were made with (tools/make_golden_models.py);
    only the classifier code around it is under test here."""
    import torch
    import torch.nn as nn

    class Fake(nn.Module):
        def __init__(self):
            super().__init__()
            self.logit_scale = nn.Parameter(torch.tensor(float(np.log(100.))))
            self.table = nn.Parameter(table)
            self.visual = types.SimpleNamespace(output_dim=C)

        def encode_image(self, imgs):
            return imgs.flatten(1)[:, :C] * 1.5

        def encode_text(self, tokens):
            return self.table[tokens[:, 0].long()]

    return Fake()


def test_adapter_kernel_matches_reference_fixture(hip):
    import torch
    from eventclip_amd.adapter import TransformerAdapter
    z = load('adapter_small.npz')
    ad = TransformerAdapter(in_dim=int(z['cfg_in_dim']), d_model=int(z['cfg_d_model']),
                            num_heads=int(z['cfg_num_heads']), ffn_dim=int(z['cfg_ffn_dim']),
                            num_layers=int(z['cfg_num_layers']), residual=float(z['cfg_residual']))
    ad.load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('w:')})
    ad = ad.cuda().eval()
    for T in z['Ts']:
        feats = torch.from_numpy(z[f'feats_T{T}']).cuda()
        valid = torch.from_numpy(z[f'valid_T{T}']).cuda()
        got = ad(feats, valid).cpu()
        want = torch.from_numpy(z[f'out_T{T}'])
        m = valid.cpu()[..., None].float()
        torch.testing.assert_close(got * m, want * m, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('T', [1, 5, 10])
def test_adapter_kernel_full_size_matches_oracle(T, hip):
    import torch
    from eventclip_amd.adapter import TransformerAdapter
    from oracle import adapter as oa
    torch.manual_seed(T)
    ad = TransformerAdapter(in_dim=768, d_model=256, num_heads=4, ffn_dim=1024, num_layers=2,
                            residual=0.95).eval()
    with torch.no_grad():
        for p in ad.parameters():
            p.add_(torch.randn_like(p) * 0.05)
    B = 9
    valid = torch.rand(B, T) < 0.6
    valid[:, 0] = True
    feats = torch.randn(B, T, 768) * valid[..., None]
    want = oa.transformer_adapter(ad.state_dict(), feats, valid, 4, 0.95)
    got = ad.cuda()(feats.cuda(), valid.cuda()).cpu()
    m = valid[..., None].float()
    torch.testing.assert_close(got * m, want * m, rtol=1e-4, atol=1e-5)


def test_zero_shot_classifier_matches_reference_fixture(hip):
    import torch
    from eventclip_amd.clip_cls import ZSCLIPClassifier
    z = load('classify_zs.npz')
    C, K = int(z['C']), int(z['K'])
    names = [f'class_{i}' for i in range(K)]
    imgs, valid = torch.from_numpy(z['imgs']).cuda(), torch.from_numpy(z['valid']).cuda()
    for agg in ('sum', 'mean'):
        model = ZSCLIPClassifier(clip_dict=dict(
            clip_model=fake_clip(C, torch.from_numpy(z['table'])), prompt='a point cloud image of a {}',
            class_names=names, agg_func=agg, class_tokens=torch.from_numpy(z['tokens']))).cuda().eval()
        o = model({'img': imgs, 'valid_mask': valid})
        assert set(o) == {'full_logits', 'valid_masks', 'logits', 'probs'}
        for k in ('full_logits', 'logits', 'probs'):
            torch.testing.assert_close(o[k].cpu(), torch.from_numpy(z[f'{agg}_{k}']), rtol=1e-5,
                                       atol=2e-4)
        assert model.state_dict() == {}            # frozen CLIP weights stay out (clip_cls.py:208)


def test_few_shot_classifier_matches_reference_fixture(hip):
    import torch
    from eventclip_amd.clip_cls import FSCLIPClassifier
    z = load('classify_fs.npz')
    C, K = int(z['C']), int(z['K'])
    names = [f'class_{i}' for i in range(K)]
    imgs, valid = torch.from_numpy(z['imgs']).cuda(), torch.from_numpy(z['valid']).cuda()
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('w:')}
    ad = dict(adapter_type='text-trans', in_dim=C, d_model=int(z['adcfg_d_model']),
              num_heads=int(z['adcfg_num_heads']), ffn_dim=int(z['adcfg_ffn_dim']),
              norm_first=True, num_layers=int(z['adcfg_num_layers']),
              residual=float(z['adcfg_residual']))
    for agg in ('sum', 'mean'):
        model = FSCLIPClassifier(
            adapter_dict=dict(ad),
            clip_dict=dict(clip_model=fake_clip(C, torch.from_numpy(z['table'])),
                           prompt='a point cloud image of a {}', class_names=names, agg_func=agg,
                           class_tokens=torch.from_numpy(z['tokens'])),
            loss_dict=dict(use_logits_loss=True, use_probs_loss=False))
        assert sorted(model.state_dict()) == sorted(sd)       # the reference's 29 keys
        model.load_state_dict(sd)
        model = model.cuda().eval()
        o = model({'img': imgs, 'valid_mask': valid})
        for k in ('full_logits', 'logits', 'probs'):
            torch.testing.assert_close(o[k].cpu(), torch.from_numpy(z[f'{agg}_{k}']), rtol=1e-4,
                                       atol=2e-4)


def test_end_to_end_events_to_logits_matches_oracle(hip):
    """events -> frames -> preprocess -> ViT-B/32 -> logits, both batch layouts, against the
    oracle chain (C events oracle, Pillow-pinned preprocess, fp32 torch CLIP, clip_cls tail)."""
    import torch
    from eventclip_amd import clip as eclip
    from eventclip_amd.clip_cls import ZSCLIPClassifier
    from eventclip_amd.event2img import Event2ImagePipeline
    from eventclip_amd.synthetic import make_batch
    from oracle import classify as oc
    from oracle import clip_ref
    from oracle import events as oe
    from oracle import preprocess as op
    qa = dict(max_imgs=3, N=20000, split_method='event_count', convert_method='event_histogram',
              grayscale=True, count_non_zero=False, background_mask=True)
    cfg = eclip.arch_config('ViT-B/32', layers=2, text_layers=2, vocab_size=49408)
    sd = eclip.random_state_dict(cfg, seed=21)
    clip_model = eclip.CLIP(cfg, sd, dtype='float16').cuda().eval()
    K = 11
    tokens = eclip.synthetic_tokens(K, seed=2)
    evs = make_batch(3, [50000, 9000, 31000], (180, 240), seed=8)
    model = ZSCLIPClassifier(clip_dict=dict(clip_model=clip_model, prompt='a {}',
                                            class_names=[str(i) for i in range(K)],
                                            agg_func='mean', class_tokens=tokens)).cuda().eval()
    fused = Event2ImagePipeline((180, 240), 225000, qa, n_px=224, patch=32, kpad=clip_model.kpad,
                                dtype=torch.float16)
    plain = Event2ImagePipeline((180, 240), 225000, qa, n_px=224)
    o1 = model(fused(evs))
    batch = plain(evs)
    assert tuple(batch['img'].shape) == (3, 3, 3, 224, 224)
    o2 = model(batch)
    # oracle chain
    frames, valid = [], torch.zeros(3, 3, dtype=torch.bool)
    for b, ev in enumerate(evs):
        f = oe.events2frames(ev, 'event_count', 'event_histogram', shape=(180, 240),
                             **{k: v for k, v in qa.items() if k not in
                                ('max_imgs', 'split_method', 'convert_method')})
        valid[b, :len(f)] = True
        frames.append(f)
    assert valid.tolist() == [[True, True, False], [True, False, False], [True, True, False]]
    imgs = torch.from_numpy(op.preprocess(np.concatenate(frames), 224))
    feats = clip_ref.encode_image(sd, cfg, imgs)
    text = torch.nn.functional.normalize(clip_ref.encode_text(sd, cfg, tokens), dim=-1)
    want = oc.zs_forward(feats, valid, text, 100.0, 'mean')
    assert torch.equal(o1['valid_masks'].cpu(), valid) and torch.equal(o2['valid_masks'].cpu(), valid)
    # image features: north_star's 1e-3 relative (max |diff| / max |ref|), measured 5e-4
    gf = clip_model.encode_image(imgs.cuda()).cpu()
    assert float((gf - feats).abs().max() / feats.abs().max()) < 1e-3
    # logits = 100 * feats . text: north_star's 1e-3 relative.  (Measured 2.5e-4 .. 5.3e-4 over
    # seeds and architectures, tools/precision_probe.py; 1.3e-3 before the patch embedding and
    # ln_post @ proj kept their lo parts.)
    mag = float(want['full_logits'].abs().max())
    for o in (o1, o2):
        for k in ('full_logits', 'logits'):
            assert float((o[k].cpu() - want[k]).abs().max()) / mag < 1e-3
        assert float((o['probs'].cpu() - want['probs']).abs().max()) < 5e-2
        assert torch.equal(o['logits'].argmax(-1).cpu(), want['logits'].argmax(-1))
    torch.testing.assert_close(o1['logits'], o2['logits'], rtol=0, atol=0)  # same kernels, same bits


def _ft_fake_clip(C, z):
    """The stand-in of tools/make_golden_ft.py with PLAIN attention blocks (what a served model has)."""
    import torch
    import torch.nn as nn

    class Block(nn.Module):
        def __init__(self, d, heads):
            super().__init__()
            self.attn = nn.MultiheadAttention(d, heads)
            self.ln_1 = nn.LayerNorm(d)

    class Visual(nn.Module):
        def __init__(self, d=16, heads=2, layers=2, out=8):
            super().__init__()
            self.conv1 = nn.Conv2d(3, d, 2, 2, bias=False)
            self.class_embedding = nn.Parameter(torch.zeros(d))
            self.transformer = nn.Module()
            self.transformer.resblocks = nn.Sequential(*[Block(d, heads) for _ in range(layers)])
            self.ln_post = nn.LayerNorm(d)
            self.proj = nn.Parameter(torch.zeros(d, out))
            self.output_dim = out

        def forward(self, x):
            for blk in self.transformer.resblocks:
                h = blk.ln_1(x)
                x = x + blk.attn(h, h, h, need_weights=False)[0]
            return self.ln_post(x[0]) @ self.proj

    class Fake(nn.Module):
        def __init__(self):
            super().__init__()
            self.logit_scale = nn.Parameter(torch.tensor(float(np.log(100.))))
            self.table = nn.Parameter(torch.from_numpy(z['table']))
            self.visual = Visual(out=C)

        def encode_image(self, imgs):
            return imgs.flatten(1)[:, :C] * 1.5

        def encode_text(self, tokens):
            return self.table[tokens[:, 0].long()]

    m = Fake()
    m.load_state_dict({k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('base:')})
    return m


@pytest.mark.parametrize('tag', ['full', 'lora'])
def test_finetuned_classifier_serves_reference_checkpoints(tag, hip):
    """FTCLIPClassifier (models/clip_cls_ft.py) for serving: a checkpoint exactly as the reference's class
    writes it -- fully fine-tuned, or with LoRA factors in every attention block -- loads key for key,
    the LoRA factors fold into plain weights that reproduce the reference's injected tower on a probe,
    and the forward (normalised features against the tuned text features) matches the reference's."""
    import torch
    from eventclip_amd.clip_cls import build_model
    from eventclip_amd.clip_cls_ft import FTCLIPClassifier
    z = load('classify_ft.npz')
    C, K = int(z['C']), int(z['K'])
    names = [f'class_{i}' for i in range(K)]
    imgs, valid = torch.from_numpy(z['imgs']).cuda(), torch.from_numpy(z['valid']).cuda()
    sd = {k[len(tag) + 4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(tag + '/sd:')}
    assert any('lora_' in k for k in sd) == (tag == 'lora')
    for agg in ('sum', 'mean'):
        params = types.SimpleNamespace(
            model='FTCLIP', adapter_dict=dict(adapter_type='text-identity', residual=True),
            clip_dict=dict(clip_model=_ft_fake_clip(C, z), prompt='a point cloud image of a {}',
                           class_names=names, agg_func=agg, class_tokens=torch.from_numpy(z['tokens']),
                           lora=-1 if tag == 'full' else 'qkvo-2', only_conv1=False, only_bias=False,
                           only_ln=False),
            loss_dict=dict(use_logits_loss=True, use_probs_loss=False))
        model = build_model(params)
        assert isinstance(model, FTCLIPClassifier)
        model.load_state_dict(sd)
        model = model.cuda().eval()
        o = model({'img': imgs, 'valid_mask': valid})
        for k in ('full_logits', 'logits', 'probs'):
            torch.testing.assert_close(o[k].cpu(), torch.from_numpy(z[f'{tag}/{agg}_{k}']), rtol=1e-4, atol=2e-4)
        # the tower the checkpoint describes: plain weights after the fold == the reference's (injected) one
        with torch.no_grad():
            got = model.model.visual.cpu()(torch.from_numpy(z['probe']))
        torch.testing.assert_close(got, torch.from_numpy(z[f'{tag}/visual_out']), rtol=1e-5, atol=1e-5)
        # what it writes back is a plain fine-tuned checkpoint: model.visual.* + text_feats + adapter.dummy
        keys = set(model.state_dict())
        assert 'text_feats' in keys and 'adapter.dummy' in keys
        assert {k for k in keys if k.startswith('model.')} == {'model.visual.' + k for k in model.model.visual.state_dict()}
        assert model.train().training and not model.model.training
        model.eval()


def test_finetuned_checkpoint_through_the_real_tower(hip):
    """A fine-tuned `model.visual.*` checkpoint swaps the HIP tower's weights: after load_state_dict the
    features are those of a CLIP built from the fine-tuned weights directly."""
    import torch
    from eventclip_amd import clip as eclip
    from eventclip_amd.clip_cls_ft import FTCLIPClassifier
    cfg = eclip.arch_config('ViT-B/32', layers=2, text_layers=1, vocab_size=1024)
    sd0, sd1 = eclip.random_state_dict(cfg, seed=1), eclip.random_state_dict(cfg, seed=2)
    tuned = dict(sd0)
    tuned.update({k: v for k, v in sd1.items() if k.startswith('visual.')})     # "fine-tuned" tower
    tokens = eclip.synthetic_tokens(5, seed=0)
    clf = FTCLIPClassifier(clip_dict=dict(clip_model=eclip.CLIP(cfg, sd0).cuda().eval(), prompt='a {}',
                                          class_names=list('abcde'), agg_func='mean', class_tokens=tokens))
    ckpt = {'model.visual.' + k[len('visual.'):]: v for k, v in sd1.items() if k.startswith('visual.')}
    ckpt['text_feats'] = torch.randn(5, cfg['embed_dim'])
    ckpt['adapter.dummy'] = torch.zeros(1)
    clf.load_state_dict(ckpt)
    clf = clf.cuda().eval()
    img = torch.randn(3, 3, 224, 224, generator=torch.Generator().manual_seed(4)).cuda()
    want = eclip.CLIP(cfg, tuned).cuda().eval().encode_image(img)
    assert torch.equal(clf.model.encode_image(img), want)
    out = clf({'img': img[None], 'valid_mask': torch.ones(1, 3, dtype=torch.bool).cuda()})
    t = torch.nn.functional.normalize(ckpt['text_feats'], dim=-1).cuda()
    f = torch.nn.functional.normalize(want, dim=-1)
    torch.testing.assert_close(out['full_logits'][0], clf.logit_scale * f @ t.T, rtol=1e-4, atol=1e-3)
