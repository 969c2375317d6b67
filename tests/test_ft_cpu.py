"""Host side of serving fine-tuned checkpoints: FTCLIPClassifier's checkpoint handling on the fixture the
reference's own class wrote (tools/make_golden_ft.py).  No GPU: the stand-in encoder is plain torch; the
forward itself is a GPU test (tests/test_models_gpu.py)."""
import types

import pytest
import torch

from test_models_gpu import _ft_fake_clip, load


@pytest.mark.parametrize('tag', ['full', 'lora'])
def test_reference_checkpoint_loads_folds_and_round_trips(tag):
    from eventclip_amd.clip_cls import build_model
    from eventclip_amd.clip_cls_ft import FTCLIPClassifier
    z = load('classify_ft.npz')
    C, K = int(z['C']), int(z['K'])
    sd = {k[len(tag) + 4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(tag + '/sd:')}
    params = types.SimpleNamespace(
        model='FTCLIP', adapter_dict=dict(adapter_type='text-identity', residual=True),
        clip_dict=dict(clip_model=_ft_fake_clip(C, z), prompt='a point cloud image of a {}',
                       class_names=[f'class_{i}' for i in range(K)], agg_func='sum',
                       class_tokens=torch.from_numpy(z['tokens']), lora=-1),
        loss_dict=dict(use_logits_loss=True, use_probs_loss=False))
    model = build_model(params)
    assert isinstance(model, FTCLIPClassifier)
    model.load_state_dict(sd)
    with torch.no_grad():
        got = model.model.visual(torch.from_numpy(z['probe']))
    torch.testing.assert_close(got, torch.from_numpy(z[f'{tag}/visual_out']), rtol=1e-6, atol=1e-6)
    assert torch.equal(model.text_feats.detach(), sd['text_feats'])
    back = model.state_dict()
    assert {k for k in back if not k.startswith('model.')} == {'text_feats', 'adapter.dummy'}
    assert all(k.startswith('model.visual.') for k in back if k.startswith('model.'))
    # a plain checkpoint written by this class loads again
    model2 = build_model(params)
    model2.load_state_dict(back)
    with torch.no_grad():
        torch.testing.assert_close(model2.model.visual(torch.from_numpy(z['probe'])), got)
    assert model.train() is model and model.training and not model.model.training   # clip_cls_ft.py:300-306
    model.eval()
    with pytest.raises(KeyError):
        build_model(params).load_state_dict({'text_feats': sd['text_feats']})
    with pytest.raises(AssertionError):
        FTCLIPClassifier(adapter_dict=dict(adapter_type='text-trans'), clip_dict=params.clip_dict)
