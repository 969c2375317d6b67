"""The fine-tuning oracle (oracle/ft_train.py) against the reference's own FTCLIPClassifier under autograd
(tests/golden/ft_train.npz, tools/make_golden_ft_train.py): which tensors train, LoRA key names, loss,
every gradient, and the parameters after two Adam steps with the two learning rates of method.py:152-186."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import ft_train as oft
from oracle import train as otrain

FLAGS = ('only_conv1', 'only_bias', 'only_ln', 'only_cls_fc', 'only_cls_token')


def golden():
    return np.load(os.path.join(GOLDEN, 'ft_train.npz'))


def case(z, tag):
    """-> dict with the case's classifier state dict, config and expectations."""
    cfg_v = z['cfg'].tolist()
    cfg = dict(image_size=cfg_v[0], patch=cfg_v[1], width=cfg_v[2], layers=cfg_v[3], heads=cfg_v[4], embed_dim=cfg_v[5])
    sd = {}
    for k in z[f'{tag}/sd_keys'].tolist():
        key = f'{tag}/sd:{k}' if f'{tag}/sd:{k}' in z.files else f'base/sd:{k}'
        sd[k] = torch.from_numpy(z[key])
    lora = str(z[f'{tag}/lora'])
    lora = lora if '-' in lora and not lora.lstrip('-').isdigit() else int(lora)
    clip_dict = dict(lora=lora, **{f: bool(v) for f, v in zip(FLAGS, z[f'{tag}/flags'].tolist())})
    visual = {k[len('model.visual.'):]: v for k, v in sd.items() if k.startswith('model.visual.')}
    prompt = str(z[f'{tag}/adapter_type']).startswith('text-')
    text = sd['text_feats'] if prompt else torch.from_numpy(z[f'{tag}/text_fixed'])
    return dict(cfg=cfg, sd=sd, visual=visual, clip_dict=clip_dict, prompt=prompt, text=text,
                agg=str(z[f'{tag}/agg']), probs_loss=bool(z[f'{tag}/probs_loss']),
                imgs=torch.from_numpy(z['imgs']), valid=torch.from_numpy(z['valid']),
                labels=torch.from_numpy(z['labels']), trainable=z[f'{tag}/trainable'].tolist())


CASES = golden()['cases'].tolist()


@pytest.mark.parametrize('tag', CASES)
def test_trainable_set_matches_the_reference(tag):
    z = golden()
    c = case(z, tag)
    names = {'model.visual.' + n for n in oft.trainable_names(c['visual'], c['clip_dict'])}
    if c['prompt']:
        names.add('text_feats')
    assert sorted(names) == c['trainable']


def test_lora_injection_produces_the_reference_key_names():
    z = golden()
    plain = case(z, 'full')['visual']
    for tag in ('lora_qkvo', 'lora_int', 'lora_qv'):
        c = case(z, tag)
        mine = oft.inject_lora(plain, c['clip_dict']['lora'], torch.Generator().manual_seed(0))
        assert sorted(mine) == sorted(c['visual'])
        for k, v in mine.items():
            assert tuple(v.shape) == tuple(c['visual'][k].shape), k
            if 'lora_up' in k:
                assert float(v.abs().max()) == 0.0           # lora.py:10
        r = oft.parse_lora(c['clip_dict']['lora'])[0]
        downs = torch.cat([v.flatten() for k, v in mine.items() if 'lora_down' in k])
        assert abs(float(downs.std()) * r - 1.0) < 0.2       # lora.py:9: std 1 / r


@pytest.mark.parametrize('tag', CASES)
def test_loss_and_gradients_match_the_reference(tag):
    z = golden()
    c = case(z, tag)
    train = oft.trainable_names(c['visual'], c['clip_dict'])
    loss, grads, out, feats = oft.loss_and_grads(c['visual'], c['cfg'], c['imgs'], c['valid'], c['labels'], c['text'],
                                                 100.0, c['agg'], c['probs_loss'], train=train,
                                                 text_trainable=c['prompt'])
    assert abs(loss - float(z[f'{tag}/loss'])) < 2e-4 * max(1.0, abs(loss))
    np.testing.assert_allclose(feats.numpy(), z[f'{tag}/feats'], rtol=2e-4, atol=2e-5)
    for k in ('full_logits', 'logits', 'probs'):
        np.testing.assert_allclose(out[k].numpy(), z[f'{tag}/{k}'], rtol=1e-3, atol=1e-4)
    checked = 0
    for name in c['trainable']:
        want = z[f'{tag}/grad:{name}']
        got = grads['text_feats' if name == 'text_feats' else name[len('model.visual.'):]].numpy()
        scale = max(float(np.abs(want).max()), 1e-6)
        assert np.abs(got - want).max() <= 2e-3 * scale, name
        checked += 1
    assert checked == len(c['trainable']) and checked == len(grads)


@pytest.mark.parametrize('tag', ['lora_qkvo', 'bias', 'conv_cls'])
def test_two_adam_steps_match_the_reference(tag):
    z = golden()
    c = case(z, tag)
    train = oft.trainable_names(c['visual'], c['clip_dict'])
    params = {k: v.double().numpy().copy() for k, v in c['visual'].items()}
    text = c['text'].double().numpy().copy()
    lr, clip_lr = float(z['lr']), float(z['clip_lr'])
    state = {}
    for step in (1, 2):
        vis = {k: torch.from_numpy(v) for k, v in params.items()}
        _, grads, _, _ = oft.loss_and_grads(vis, c['cfg'], c['imgs'], c['valid'], c['labels'], torch.from_numpy(text),
                                            100.0, c['agg'], c['probs_loss'], train=train, text_trainable=c['prompt'])
        for name, g in grads.items():
            p = text if name == 'text_feats' else params[name]
            m, v = state.setdefault(name, (np.zeros_like(p), np.zeros_like(p)))
            otrain.adam_step(p, g.double().numpy(), m, v, step, lr if name == 'text_feats' else clip_lr)
    for name in c['trainable']:
        want = z[f'{tag}/step2:{name}']
        got = text if name == 'text_feats' else params[name[len('model.visual.'):]]
        # Adam moves an element by ~lr whatever the size of its gradient: where the gradient is zero in exact
        # arithmetic (the key bias: softmax does not see it) the reference's update follows fp32 noise
        g0 = np.abs(z[f'{tag}/grad:{name}'])
        live = g0 > 1e-4 * g0.max()
        assert live.any(), name
        np.testing.assert_allclose(got[live], want[live], rtol=0, atol=2e-4 * max(1.0, float(np.abs(want).max())),
                                   err_msg=name)
