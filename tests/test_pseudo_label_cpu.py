"""The pseudo-label oracle on a hand-worked example of gen_data.py:132-164 / :196-215."""
import torch

from oracle import pseudo_label as opl


def test_hand_worked_tta_example():
    # two samples x four views x three classes
    p = torch.tensor([
        [.7, .2, .1], [.6, .3, .1], [.8, .1, .1], [.5, .4, .1],     # all views say class 0, min top = .5
        [.1, .6, .3], [.1, .5, .4], [.2, .3, .5], [.1, .7, .2],     # third view dissents (class 2)
    ])
    r = opl.select(p, 0.45, tta=True)
    assert r['pred'].tolist() == [0, 1]
    torch.testing.assert_close(r['max_prob'], torch.tensor([.65, .525]))
    assert r['selected'].tolist() == [True, True]
    assert opl.select(p, 0.45, True, tta_consistent=True)['selected'].tolist() == [True, False]
    assert opl.select(p, 0.55, True, tta_min_prob=True)['selected'].tolist() == [False, False]
    assert opl.select(p, 0.45, True, tta_min_prob=True)['selected'].tolist() == [True, True]
    # without TTA every row is its own sample
    r1 = opl.select(p, 0.55)
    assert r1['selected'].tolist() == [True, True, True, False, True, False, False, True]


def test_topk_keeps_most_confident_per_class():
    pred = torch.tensor([0, 0, 0, 1, 1, 2])
    prob = torch.tensor([.9, .5, .7, .6, .8, .4])
    sel = torch.tensor([True, True, True, True, False, True])
    assert opl.topk_per_class(pred, prob, sel, 3, 2).tolist() == [True, False, True, True, False, True]
    assert opl.topk_per_class(pred, prob, sel, 3, 1).tolist() == [True, False, False, True, False, True]


def _golden():
    import os
    import numpy as np
    from conftest import GOLDEN
    return np.load(os.path.join(GOLDEN, 'pseudo_label.npz'))


def _parse(tag):
    parts = tag.split('_')
    return (parts[0] == 'tta1', parts[1] == 'c1', parts[2] == 'm1', float(parts[3][1:]), int(parts[4][1:]))


def test_oracle_matches_the_references_gen_data_main():
    """tests/golden/pseudo_label.npz: per-sample pseudo-labels written by the reference's own gen_data.main()
    (tools/make_golden_pseudo.py drives it with stand-ins around the selection code) for 30 flag combinations."""
    import numpy as np
    z = _golden()
    probs4, labels, K = torch.from_numpy(z['probs4']), torch.from_numpy(z['labels']), int(z['K'])
    for tag in z['cases']:
        tta, cons, minp, thr, topk = _parse(str(tag))
        p = probs4.flatten(0, 1) if tta else probs4[:, 0]
        r = opl.select(p, thr, tta, cons, minp)
        keep = opl.topk_per_class(r['pred'], r['max_prob'], r['selected'], K, topk) if topk > 0 else r['selected']
        got = torch.where(keep, r['pred'], torch.full_like(r['pred'], -1)).numpy()
        np.testing.assert_array_equal(got, z['sel_' + str(tag)], err_msg=str(tag))
