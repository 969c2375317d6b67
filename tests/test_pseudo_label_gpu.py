"""ec_pseudo_label against the oracle restatement of gen_data.py:132-164 (MI355X)."""
import itertools

import pytest

pytestmark = pytest.mark.gpu


def make_probs(B, V, K, seed, sharp):
    import torch
    g = torch.Generator().manual_seed(seed)
    base = torch.randn(B, 1, K, generator=g) * sharp
    views = base + torch.randn(B, V, K, generator=g) * 0.7
    return views.softmax(-1).reshape(B * V, K)


@pytest.mark.parametrize('K', [2, 101, 1000])
@pytest.mark.parametrize('consistent,min_prob', list(itertools.product([False, True], repeat=2)))
def test_tta_selection_matches_oracle(K, consistent, min_prob, hip):
    import torch
    from eventclip_amd import pseudo_label as pl
    from oracle import pseudo_label as opl
    B = 300
    probs = make_probs(B, 4, K, seed=K + 2 * consistent + min_prob, sharp=3.0 if K > 2 else 1.0)
    for thr in (0.0, 0.3, 0.9):
        want = opl.select(probs, thr, True, consistent, min_prob)
        got = pl.select(probs.cuda(), thr, True, consistent, min_prob)
        torch.testing.assert_close(got['probs'].cpu(), want['probs'], rtol=1e-6, atol=1e-7)
        # decisions can only differ where a probability sits within rounding of the threshold
        border = (want['max_prob'] - thr).abs() < 1e-6
        assert torch.equal(got['pred'].cpu()[~border], want['pred'][~border])
        assert torch.equal(got['selected'].cpu()[~border], want['selected'][~border])
        torch.testing.assert_close(got['max_prob'].cpu(), want['max_prob'], rtol=1e-6, atol=1e-7)
        assert 0 < int(want['selected'].sum()) or thr > 0.5 or consistent or min_prob


def test_single_view_and_ties(hip):
    import torch
    from eventclip_amd import pseudo_label as pl
    from oracle import pseudo_label as opl
    probs = make_probs(257, 1, 101, seed=9, sharp=2.0)
    want = opl.select(probs, 0.25)
    got = pl.select(probs.cuda(), 0.25)
    assert torch.equal(got['probs'].cpu(), want['probs'])            # V = 1: passed through unchanged
    assert torch.equal(got['pred'].cpu(), want['pred'])
    assert torch.equal(got['selected'].cpu(), want['selected'])
    # exact ties go to the lowest class index, like torch.max / torch.argmax
    t = torch.zeros(4 * 2, 5)
    t[:, 3] = 0.5
    t[:, 1] = 0.5
    got = pl.select(t.cuda(), 0.4, True, True, True)
    assert got['pred'].tolist() == [1, 1] and got['selected'].tolist() == [True, True]
    # one dissenting view vetoes a sample under tta_consistent only
    t[2, 1] = 0.2
    assert pl.select(t.cuda(), 0.4, True, True, False)['selected'].tolist() == [False, True]
    assert pl.select(t.cuda(), 0.4, True, False, False)['selected'].tolist() == [True, True]


def test_topk_per_class(hip):
    import torch
    from eventclip_amd import pseudo_label as pl
    from oracle import pseudo_label as opl
    probs = make_probs(400, 4, 7, seed=4, sharp=2.0)
    got = pl.select(probs.cuda(), 0.3, True, True, False)
    for k in (0, 1, 5, 1000):
        keep = pl.topk_per_class(got['pred'], got['max_prob'], got['selected'], 7, k)
        if k == 0:
            assert torch.equal(keep, got['selected'])
            continue
        want = opl.topk_per_class(got['pred'].cpu(), got['max_prob'].cpu(), got['selected'].cpu(), 7, k)
        assert torch.equal(keep.cpu(), want)
        assert int(keep.sum()) <= 7 * k


def test_labeler_end_to_end_tiny(hip):
    """Raw events -> 4 TTA views -> classifier -> selection, against the oracle on the same probs."""
    import torch
    from eventclip_amd import clip
    from eventclip_amd.clip_cls import ZSCLIPClassifier
    from eventclip_amd.event2img import Event2ImagePipeline
    from eventclip_amd.pseudo_label import PseudoLabeler
    from eventclip_amd.synthetic import make_events
    from oracle import pseudo_label as opl
    cfg = clip.arch_config('ViT-B/32', layers=2, text_layers=2, vocab_size=49408)
    m = clip.CLIP(cfg, clip.random_state_dict(cfg, seed=3)).cuda().eval()
    K = 6
    tokens = clip.synthetic_tokens(K, seed=1)
    clf = ZSCLIPClassifier(clip_dict=dict(clip_model=m, class_names=[f'c{i}' for i in range(K)],
                                          class_tokens=tokens, prompt='a {}',
                                          agg_func='mean')).cuda().eval()
    res = (100, 120)
    qa = dict(split_method='event_count', convert_method='event_histogram', max_imgs=2, N=6000,
              grayscale=True, count_non_zero=False, background_mask=True)
    pipe = Event2ImagePipeline(res, 12000, qa, n_px=224, patch=32, kpad=m.kpad)
    samples = [make_events(n, res, seed=20 + i) for i, n in enumerate((13000, 6000, 2500))]
    lab = PseudoLabeler(clf, pipe, conf_thresh=1.0 / K, tta=True, tta_consistent=True)
    got = lab(samples)
    views = [clf(b)['probs'] for b in pipe.tta(samples)]
    want = opl.select(torch.stack(views, 1).flatten(0, 1).cpu(), 1.0 / K, True, True, False)
    assert torch.equal(got['pred'].cpu(), want['pred'])
    assert torch.equal(got['selected'].cpu(), want['selected'])


def test_kernel_matches_the_references_gen_data_main(hip):
    """ec_pseudo_label + topk_per_class against the per-sample pseudo-labels the reference's own
    gen_data.main() wrote (tests/golden/pseudo_label.npz), all 30 flag combinations."""
    import os
    import numpy as np
    import torch
    from conftest import GOLDEN
    from eventclip_amd import pseudo_label as pl
    z = np.load(os.path.join(GOLDEN, 'pseudo_label.npz'))
    probs4, K = torch.from_numpy(z['probs4']).cuda(), int(z['K'])
    for tag in z['cases']:
        parts = str(tag).split('_')
        tta, cons, minp = parts[0] == 'tta1', parts[1] == 'c1', parts[2] == 'm1'
        thr, topk = float(parts[3][1:]), int(parts[4][1:])
        p = probs4.flatten(0, 1) if tta else probs4[:, 0].contiguous()
        r = pl.select(p, thr, tta, cons, minp)
        keep = pl.topk_per_class(r['pred'], r['max_prob'], r['selected'], K, topk)
        got = torch.where(keep, r['pred'], torch.full_like(r['pred'], -1)).cpu().numpy()
        np.testing.assert_array_equal(got, z['sel_' + str(tag)], err_msg=str(tag))
