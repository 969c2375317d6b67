"""ec_randaugment on the MI355X against the Pillow-pinned oracle and the fixture the reference's own
RandAugment class produced (tools/make_golden_randaugment.py): bit for bit."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def test_every_operator_and_bin_matches_the_oracle(hip):
    import torch
    from eventclip_amd import randaugment as ra
    from eventclip_amd.synthetic import make_events
    from oracle import events as oe
    from oracle import randaugment as ora
    rng = np.random.default_rng(0)
    ev = make_events(9000, (90, 120), seed=3)
    imgs = [rng.integers(0, 256, size=(45, 60, 3), dtype=np.uint8),
            oe.events2frames(ev, shape=(90, 120), N=9000, grayscale=False, count_non_zero=False,
                             background_mask=True)[0],
            np.full((20, 33, 3), 77, dtype=np.uint8),
            rng.integers(0, 256, size=(32, 32, 3), dtype=np.uint8)]
    for img in imgs:
        H, W, _ = img.shape
        cases = []
        for op in ra.OP_NAMES:
            table = ora.magnitude_table(op, (H, W))
            mags = [0.0] if table is None else sorted({float(v) for v in table.tolist()})
            if op in ora.SIGNED:
                mags = mags + [-m for m in mags if m]
            if op == 'Rotate':
                mags += [90.0, 180.0, 270.0, 360.0]
            cases += [(op, m) for m in mags]
        # one launch: frame i gets case i as a single-op list
        frames = torch.from_numpy(np.repeat(img[None], len(cases), axis=0)).cuda()
        for fill in ((255, 255, 255), (0, 0, 0), (10, 200, 77)):
            got = ra.apply_ops(frames, [[c] for c in cases], fill).cpu().numpy()
            for i, (op, mag) in enumerate(cases):
                want = ora.apply_op(img, op, mag, fill)
                if not np.array_equal(got[i], want):
                    d = np.argwhere(got[i] != want)
                    raise AssertionError(f'{op} {mag} fill {fill} {img.shape}: {len(d)} bytes differ, first at '
                                         f'{d[0].tolist()}: {got[i][tuple(d[0])]} vs {want[tuple(d[0])]}')


def test_matches_the_references_forward_fixture(hip):
    import torch
    from eventclip_amd import randaugment as ra
    z = np.load(os.path.join(GOLDEN, 'randaugment.npz'))
    for tag in z['cases']:
        frames = torch.from_numpy(z[f'g{int(z[tag + "_geo"])}_frames_in']).cuda()
        ops = list(zip(z[tag + '_op_names'].tolist(), z[tag + '_op_mags'].tolist()))
        got = ra.apply_ops(frames, [ops] * frames.shape[0], z[tag + '_fill'].tolist())
        np.testing.assert_array_equal(got.cpu().numpy(), z[tag + '_frames_out'], err_msg=f'{tag} {ops}')
        # the class with the reference's interface, same seed -> same draws -> same frames
        aug = ra.RandAugment(num_ops=2, interpolation='bicubic', fill=z[tag + '_fill'].tolist())
        torch.manual_seed(int(z[tag + '_seed']))
        np.testing.assert_array_equal(aug(frames).cpu().numpy(), z[tag + '_frames_out'])
        assert aug.cur_ops is None


def test_three_op_chain_and_per_frame_lists(hip):
    import torch
    from eventclip_amd import randaugment as ra
    from oracle import randaugment as ora
    rng = np.random.default_rng(4)
    frames = rng.integers(0, 256, size=(5, 50, 70, 3), dtype=np.uint8)
    lists = [[('Rotate', 13.0), ('Equalize', 0.0), ('Sharpness', -0.6)],
             [('Contrast', 0.9), ('ShearY', -0.2), ('AutoContrast', 0.0)],
             [('Identity', 0.0), ('Identity', 0.0), ('Identity', 0.0)],
             [('Solarize', 100.5), ('TranslateX', 11.0), ('Color', 0.45)],
             [('Posterize', 4.0), ('Brightness', -0.9), ('TranslateY', -7.0)]]
    got = ra.apply_ops(torch.from_numpy(frames).cuda(), lists, (255, 255, 255)).cpu().numpy()
    for f, ops, g in zip(frames, lists, got):
        np.testing.assert_array_equal(g, ora.randaugment(f[None], ops, (255, 255, 255))[0], err_msg=str(ops))


def test_pipeline_augment_flag_draws_once_per_sample(hip):
    """Event2ImagePipeline(augment=True): RandAugment between events->frames and the CLIP preprocess
    (event2img.py:118-122), one draw per sample in sample order, the same ops for its views."""
    import torch
    from eventclip_amd.event2img import Event2ImagePipeline
    from eventclip_amd.preprocess import preprocess_frames
    from eventclip_amd.synthetic import make_batch
    from oracle import events as oe
    from oracle import randaugment as ora
    res = (100, 120)
    qa = dict(max_imgs=3, N=3000, split_method='event_count', convert_method='event_histogram',
              grayscale=False, count_non_zero=False, background_mask=True)
    evs = make_batch(3, [9000, 3100, 6000], res, seed=9)
    pipe = Event2ImagePipeline(res, 9000, qa, n_px=224, augment=True)
    torch.manual_seed(123)
    out = pipe(evs)
    torch.manual_seed(123)
    want = []
    for ev in evs:
        ops = ora.randomize_ops(res)
        f = oe.events2frames(ev, shape=res, N=3000, grayscale=False, count_non_zero=False, background_mask=True)
        want.append(ora.randaugment(f, ops, (255, 255, 255)))
    ref = preprocess_frames(torch.from_numpy(np.concatenate(want)).cuda(), 224, mode='chw')
    assert torch.equal(out['img'][out['valid_mask']], ref)
