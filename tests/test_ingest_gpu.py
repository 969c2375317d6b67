"""Packed 8-byte events on the device: same frames, counts and centring as the float path."""
import numpy as np
import pytest

from conftest import event_fixture_paths, load_event_fixture

pytestmark = pytest.mark.gpu


def _inside(ev, shape):
    H, W = shape
    keep = (ev[:, 0] >= 0) & (ev[:, 1] >= 0) & (ev[:, 0] < W) & (ev[:, 1] < H)
    return np.ascontiguousarray(ev[keep])


def test_device_pack_matches_host_pack(hip):
    import torch
    from eventclip_amd import vis
    for path in event_fixture_paths()[:6]:
        ev, shape, kw, exp = load_event_fixture(path)
        ev = _inside(ev, shape)
        got, bad = vis.pack_events_device(torch.from_numpy(ev).cuda(), return_bad=True)
        assert bad == 0
        np.testing.assert_array_equal(got.cpu().numpy().view(np.uint64), vis.pack_events(ev))
    # unrepresentable events are counted and neutralised (polarity code 0)
    ev = np.array([[1.5, 2, 0, 1], [3, 4, 0, -1], [-2, 1, 0, 1], [70000, 1, 0, 1]], dtype=np.float32)
    got, bad = vis.pack_events_device(torch.from_numpy(ev).cuda(), return_bad=True)
    assert bad == 3
    codes = (got.cpu().numpy().view(np.uint64) >> np.uint64(32)) & np.uint64(3)
    np.testing.assert_array_equal(codes, [0, 2, 0, 0])


@pytest.mark.parametrize('path', event_fixture_paths(), ids=lambda p: p.split('/')[-1])
def test_packed_frames_equal_float_frames_and_oracle(path, hip):
    import torch
    from eventclip_amd import vis
    from oracle import events as oe
    ev, shape, kw, exp = load_event_fixture(path)
    ev = _inside(ev, shape)
    if ev.shape[0] == 0:
        pytest.skip('no in-sensor events')
    if not (ev[:, :2] == np.trunc(ev[:, :2])).all():
        with pytest.raises(ValueError):      # fractional coordinates have no packed form
            vis.pack_events(ev)
        return
    kw = dict(kw)
    N = int(kw.pop('N'))
    idx0, idx1 = vis.chunk_bounds(ev.shape[0], N)
    rng = torch.tensor(np.stack([idx0, idx1], 1), dtype=torch.int64).cuda()
    gray = kw.pop('grayscale', True)
    args = dict(grayscale=gray, thresh=float(kw.get('thresh', 10.)), count_non_zero=kw.get('count_non_zero', False),
                background_mask=kw.get('background_mask', True), return_counts=True, max_frame_events=N)
    ev_d = torch.from_numpy(ev).cuda()
    pk_d = torch.from_numpy(vis.pack_events(ev).view(np.int64)).cuda()
    for flip_x, negate_p in ((False, False), (True, True)):
        f0, r0, k0 = vis.events_to_frames_device(ev_d, rng, shape, flip_x=flip_x, negate_p=negate_p, **args)
        f1, r1, k1 = vis.events_to_frames_device(pk_d, rng, shape, flip_x=flip_x, negate_p=negate_p, **args)
        assert torch.equal(f0, f1) and torch.equal(r0, r1) and torch.equal(k0, k1)
    want = oe.events2frames(ev, 'event_count', 'event_histogram', shape=shape, N=N, grayscale=gray,
                            **{k: v for k, v in kw.items()})
    f1 = vis.events_to_frames_device(pk_d, rng, shape, **{**args, 'return_counts': False})
    np.testing.assert_array_equal(f1.cpu().numpy(), want)


def test_packed_uncached_long_frames(hip):
    """Frames longer than the LDS event cache take the streaming path."""
    import torch
    from eventclip_amd import vis
    from eventclip_amd.synthetic import make_events
    shape = (480, 640)
    ev = make_events(3 * 70000, shape, seed=11)
    rng = torch.tensor([[0, 70000], [70000, 140000], [140000, 210000]], dtype=torch.int64).cuda()
    f0 = vis.events_to_frames_device(torch.from_numpy(ev).cuda(), rng, shape, grayscale=False, max_frame_events=70000)
    f1 = vis.events_to_frames_device(torch.from_numpy(vis.pack_events(ev).view(np.int64)).cuda(), rng, shape,
                                     grayscale=False, max_frame_events=70000)
    assert torch.equal(f0, f1)


def test_center_packed_matches_float_center(hip):
    import torch
    from eventclip_amd import vis
    from oracle import event_utils as eu
    from conftest import GOLDEN
    import os
    z = np.load(os.path.join(GOLDEN, 'event_utils.npz'))
    res = tuple(int(v) for v in z['resolution'])
    evs = [z[f'in{i}'].astype(np.float32) for i in range(int(z['n_cases']))]
    offs = np.concatenate([[0], np.cumsum([len(e) for e in evs])])
    sr = torch.tensor(np.stack([offs[:-1], offs[1:]], 1), dtype=torch.int64).cuda()
    pk = torch.from_numpy(np.concatenate([vis.pack_events(e) for e in evs]).view(np.int64)).cuda()
    vis.center_events_device(pk, sr, res)
    got = vis.unpack_events(pk.cpu().numpy().view(np.uint64))
    for i, e in enumerate(evs):
        want = z[f'center{i}']                              # the reference's own output
        g = got[offs[i]:offs[i + 1]]
        np.testing.assert_array_equal(g[:, :2], want[:, :2])
        np.testing.assert_array_equal(g[:, 3], want[:, 3])
        np.testing.assert_allclose(g[:, 2], want[:, 2], rtol=0, atol=2e-6)   # whole microseconds


def test_pipeline_accepts_packed_samples(hip):
    import torch
    from eventclip_amd.event2img import Event2ImagePipeline
    from eventclip_amd.synthetic import make_events
    from eventclip_amd import vis
    res = (180, 240)
    qa = dict(split_method='event_count', convert_method='event_histogram', max_imgs=4, N=20000,
              grayscale=False, count_non_zero=False, background_mask=True)
    pipe = Event2ImagePipeline(res, 225000, qa, n_px=224, patch=14, kpad=1216, dtype=torch.float16)
    samples = [make_events(n, res, seed=5 + i) for i, n in enumerate((45000, 20000, 9000))]
    a = pipe(samples, center=True)
    b = pipe([vis.pack_events(s) for s in samples], center=True)
    assert torch.equal(a['valid_mask'], b['valid_mask']) and torch.equal(a['row_idx'], b['row_idx'])
    assert torch.equal(a['patches'], b['patches'])
    for va, vb in zip(pipe.tta(samples), pipe.tta([vis.pack_events(s) for s in samples])):
        assert torch.equal(va['patches'], vb['patches'])


def test_event_augmentation_matches_reference_and_feeds_the_pipeline(hip):
    """ec_augment_events == NCaltech101._augment_events' own output for replayed draws (bit for bit,
    including the dropped events and the time flip), batched, and the pipeline accepts the result."""
    import os
    import torch
    from conftest import GOLDEN
    from eventclip_amd import augment, vis
    from eventclip_amd.event2img import Event2ImagePipeline
    from oracle import event_utils as eu
    z = np.load(os.path.join(GOLDEN, 'event_utils.npz'))
    res = tuple(int(v) for v in z['resolution'])
    ms = int(z['aug_max_shift'])
    for i in range(int(z['n_cases'])):
        for seed in (0, 1, 2, 3):
            for ft in (0, 1):
                np.random.seed(1000 * i + 10 * seed + ft)
                got = augment.augment_events(z[f'in{i}'], res, ms, bool(ft))
                np.testing.assert_array_equal(got, z[f'aug{i}_{seed}_{ft}'])
    # a batch: per-sample parameters, survivors at the samples' original offsets
    evs = [z[f'in{i}'].astype(np.float32) for i in range(int(z['n_cases']))]
    prm = np.array([[5, -7, 1, 0], [-12, 12, 0, 1], [0, 0, 1, 1]], dtype=np.int32)
    cat = torch.from_numpy(np.concatenate(evs)).cuda()
    out, counts, starts = augment.augment_events_device(cat, [len(e) for e in evs], prm, res)
    for b, e in enumerate(evs):
        want = eu.augment_events(e, prm[b], res)
        assert counts[b] == len(want)
        np.testing.assert_array_equal(out[starts[b]:starts[b] + counts[b]].cpu().numpy(), want)
    # frames of the augmented batch == frames of the separately augmented samples
    qa = dict(split_method='event_count', convert_method='event_histogram', max_imgs=3, N=200,
              grayscale=True, count_non_zero=False, background_mask=True)
    pipe = Event2ImagePipeline(res, 600, qa, n_px=224, patch=32, kpad=6144)
    a = pipe(out, counts, starts=starts)
    b_ = pipe([eu.augment_events(e, prm[b], res) for b, e in enumerate(evs)])
    assert torch.equal(a['valid_mask'], b_['valid_mask']) and torch.equal(a['patches'], b_['patches'])
