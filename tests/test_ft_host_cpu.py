"""Host-side rules of the fine-tuning path (no GPU): which tensors train, LoRA specs, the gradient scaler's
policy, and that the module refuses to run without the HIP path."""
import pytest
import torch

from eventclip_amd import ft


def names(layers=2):
    out = ['class_embedding', 'positional_embedding', 'proj', 'conv1.weight', 'ln_pre.weight', 'ln_pre.bias']
    for i in range(layers):
        p = f'transformer.resblocks.{i}.'
        out += [p + k for k in ('ln_1.weight', 'ln_1.bias', 'attn.in_proj_weight', 'attn.in_proj_bias',
                                'attn.out_proj.weight', 'attn.out_proj.bias', 'ln_2.weight', 'ln_2.bias',
                                'mlp.c_fc.weight', 'mlp.c_fc.bias', 'mlp.c_proj.weight', 'mlp.c_proj.bias')]
    return out + ['ln_post.weight', 'ln_post.bias']


def test_trainable_sets_follow_the_reference_rules():
    """The same answers as oracle/ft_train.py (pinned against the reference's _build_clip, clip_cls_ft.py:44-80)."""
    from oracle import ft_train as oft
    ns = names()
    sd = {n: torch.zeros(1) for n in ns}
    for cd in (dict(lora=-1), dict(lora=-1, only_bias=True), dict(lora=-1, only_ln=True),
               dict(lora=-1, only_conv1=True, only_cls_fc=True), dict(lora=-1, only_cls_token=True),
               dict(lora='qkvo-4'), dict(lora=8, only_bias=True), dict(lora='qv-2', only_ln=True, only_cls_token=True)):
        mine = set(ft.trainable_visual(ns, cd))
        ref = set(oft.trainable_names(sd, cd))            # (plain keys: the LoRA factors are not module parameters here)
        assert mine == ref, cd
    assert ft.trainable_visual(ns, dict(lora=-1)) == ns          # order preserved: the flat buffer's layout
    assert ft.trainable_visual(ns, dict(lora='qkvo-4')) == []    # LoRA alone: only the factors train


def test_lora_spec_parsing():
    assert ft.parse_lora(-1) is None and ft.parse_lora(0) is None and ft.parse_lora(None) is None
    assert ft.parse_lora(16) == (16, True, False)                # lora.py:362-364
    assert ft.parse_lora('qv-4') == (4, False, False)
    assert ft.parse_lora('qkv-8') == (8, True, False)
    assert ft.parse_lora('qkvo-16') == (16, True, True)
    with pytest.raises(AssertionError):
        ft.parse_lora('kv-4')                                    # lora.py:353: q and v are mandatory


def test_grad_scaler_policy_is_torchs():
    s = ft.GradScaler(init_scale=1024.0, growth_interval=3)
    for found in (False, False, True, False, False, False, False):
        s.update(found)
    # two clean steps, an overflow (halve, restart the count), three clean steps (double), one more
    assert s.scale == 1024.0 and s._good == 1
    off = ft.GradScaler(enabled=False)
    off.update(True)
    assert off.scale == 1.0


def test_cosine_warmup_matches_the_few_shot_trainers_schedule():
    from eventclip_amd.train import cosine_warmup_lr
    lrs = [cosine_warmup_lr(s, 100, 2e-5, 2e-7, 5) for s in range(100)]
    assert lrs[0] == 2e-7 and abs(lrs[5] - 2e-5) < 1e-12 and lrs[99] < 3e-7
    assert all(a <= b for a, b in zip(lrs[:5], lrs[1:6])) and all(a >= b for a, b in zip(lrs[5:-1], lrs[6:]))


def test_no_cpu_fallback():
    from eventclip_amd import _lib
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(_lib.HipLibraryError):
        ft.ft_loss_grad(torch.zeros(1, 1, 4), torch.ones(1, 1, dtype=torch.bool), torch.zeros(1, dtype=torch.long),
                        torch.zeros(2, 4), 100.0)
    with pytest.raises(_lib.HipLibraryError):
        ft.VisualTower(object())
