"""HIP events->frames kernel against the oracle and the reference's golden vectors (MI355X)."""
import hashlib
import os

import numpy as np
import pytest

from conftest import event_fixture_paths, load_event_fixture

pytestmark = pytest.mark.gpu


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def run_hip(ev, shape, kw, max_frame_events=0, float_stage='float64'):
    import torch
    from eventclip_amd import vis
    idx0, idx1 = vis.chunk_bounds(ev.shape[0], kw['N'])
    ev_d = torch.from_numpy(np.ascontiguousarray(ev, dtype=np.float32)).cuda()
    rng = torch.tensor(np.stack([idx0, idx1], 1), dtype=torch.int64).cuda()
    frames, raw, kept, stats = vis.events_to_frames_device(
        ev_d, rng, shape, grayscale=kw['grayscale'], count_non_zero=kw['count_non_zero'],
        background_mask=kw['background_mask'], return_counts=True, return_stats=True,
        max_frame_events=max_frame_events, float_stage=float_stage)
    torch.cuda.synchronize()
    return frames.cpu().numpy(), raw.cpu().numpy(), kept.cpu().numpy(), stats


def run_hip_frames_only(ev, shape, kw, max_frame_events, float_stage):
    """The PRODUCT form of the launch: uint8 frames only, no debug outputs -- with them set the persistent kernels
    take their debug branches (events.hip: the row-band kernel is disabled outright and the whole-frame kernel keeps
    the old survivor pass), so the reference's fixtures would never reach the code bench.py times."""
    import torch
    from eventclip_amd import vis
    idx0, idx1 = vis.chunk_bounds(ev.shape[0], kw['N'])
    ev_d = torch.from_numpy(np.ascontiguousarray(ev, dtype=np.float32)).cuda()
    rng = torch.tensor(np.stack([idx0, idx1], 1), dtype=torch.int64).cuda()
    frames = vis.events_to_frames_device(ev_d, rng, shape, grayscale=kw['grayscale'],
                                         count_non_zero=kw['count_non_zero'], background_mask=kw['background_mask'],
                                         max_frame_events=max_frame_events, float_stage=float_stage)
    torch.cuda.synchronize()
    assert isinstance(frames, torch.Tensor) and frames.dtype == torch.uint8
    return frames.cpu().numpy()


@pytest.mark.parametrize('path', event_fixture_paths(), ids=os.path.basename)
def test_frames_only_product_kernels_match_reference_fixture(path, hip):
    """Reference vis.py:6-41 through the frames-only launch (events_pack10_kernel's lean pass for the small sensors,
    events_band10_kernel for the 480 x 640 N-ImageNet fixtures): the uint8 frames the reference's own
    make_event_histogram produced, both float stages, by hash."""
    from eventclip_amd import vis
    ev, shape, kw, exp = load_event_fixture(path)
    idx0, idx1 = vis.chunk_bounds(ev.shape[0], kw['N'])
    nmax = max(b - a for a, b in zip(idx0, idx1))
    for stage, key in (('float64', 'frames_sha256'), ('float32', 'frames_f32_sha256')):
        for cache in (nmax, 0):      # with and without the on-chip event cache
            frames = run_hip_frames_only(ev, shape, kw, cache, stage)
            assert frames.shape[0] == exp['n_frames']
            assert sha(frames) == exp[key], (os.path.basename(path), stage, cache)


@pytest.mark.parametrize('path', event_fixture_paths(), ids=os.path.basename)
def test_hip_matches_reference_fixture(path, hip):
    from oracle import events as oe
    ev, shape, kw, exp = load_event_fixture(path)
    frames, raw, kept, stats = run_hip(ev, shape, kw)
    # same frames through the LDS event cache (HBM reads each event once)
    nmax = max(b - a for a, b in zip(*__import__('eventclip_amd.vis', fromlist=['x']).chunk_bounds(
        ev.shape[0], kw['N'])))
    c_frames, c_raw, c_kept, c_stats = run_hip(ev, shape, kw, max_frame_events=nmax)
    np.testing.assert_array_equal(c_frames, frames)
    np.testing.assert_array_equal(c_raw, raw)
    np.testing.assert_array_equal(c_kept, kept)
    o_frames, o_raw, o_kept = oe.events2frames(ev, 'event_count', 'event_histogram', shape=shape,
                                               return_counts=True, float_stage='float64', **kw)
    assert frames.shape == o_frames.shape and frames.shape[0] == exp['n_frames']
    np.testing.assert_array_equal(raw, o_raw)             # bit-exact counts
    assert sha(raw.astype(np.int32)) == exp['raw_sha256']
    assert int(stats['ambiguous'].sum()) == 0
    np.testing.assert_array_equal(kept, o_kept)           # same hot pixels removed
    np.testing.assert_array_equal(frames, o_frames)       # bit-exact uint8 frames
    assert sha(frames) == exp['frames_sha256']
    assert int(stats['dropped'].sum()) == 0
    np.testing.assert_array_equal(stats['sum'], o_raw.reshape(o_raw.shape[0], -1).sum(1))
    # float32 stage (the reference under its pinned numpy 1.25): same counts, frames as recorded
    f_frames, f_raw, f_kept, _ = run_hip(ev, shape, kw, max_frame_events=nmax, float_stage='float32')
    np.testing.assert_array_equal(f_raw, raw)
    np.testing.assert_array_equal(f_kept, kept)
    assert sha(f_frames) == exp['frames_f32_sha256']
    assert int((f_frames != frames).sum()) == exp['f32_differs']


def test_numpy_api_drop_in(hip):
    """events2frames(events, split_method, convert_method, shape, **kw) like vis.py:75."""
    from eventclip_amd import vis
    from eventclip_amd.synthetic import make_events
    from oracle import events as oe
    ev = make_events(70000, (100, 120), seed=9)
    kw = dict(N=30000, grayscale=False, count_non_zero=True, background_mask=False, max_imgs=2)
    got = vis.events2frames(ev, 'event_count', 'event_histogram', shape=(100, 120), **dict(kw))
    want = oe.events2frames(ev, 'event_count', 'event_histogram', shape=(100, 120), **dict(kw))
    assert got.dtype == np.uint8
    np.testing.assert_array_equal(got, want)
    # dict input (vis.py:46-47)
    d = dict(x=ev[:, 0], y=ev[:, 1], t=ev[:, 2], p=ev[:, 3])
    got2 = vis.events2frames(d, 'event_count', 'event_histogram', shape=(100, 120), **dict(kw))
    np.testing.assert_array_equal(got2, want)
    with pytest.raises(NotImplementedError):
        vis.events2frames(ev, 'event_count', 'voxel', shape=(100, 120), N=30000)
    with pytest.raises(AssertionError):
        vis.events2frames(ev, 'time', 'event_histogram', shape=(100, 120), N=30000)


def test_out_of_sensor_is_dropped_and_reported(hip):
    from eventclip_amd import vis
    ev = np.array([[5, 5, 0, 1], [300, 5, 0.1, 1], [-3, 2, 0.2, -1], [6, 5, 0.3, -1]], np.float32)
    with pytest.raises(ValueError):
        vis.events2frames(ev, 'event_count', 'event_histogram', shape=(36, 52), N=10)


@pytest.mark.parametrize('geom', ['n_caltech', 'n_cars', 'n_imagenet'])
def test_full_size_batch_properties(geom, hip):
    """BASELINE-sized batch: counts sum to the event count and match the oracle on a sample."""
    import torch
    from eventclip_amd import vis
    from eventclip_amd.synthetic import GEOMETRY, make_events
    from oracle import events as oe
    g = GEOMETRY[geom]
    shape, N = g['resolution'], g['N']
    B = 16
    n_ev = g['max_n'] if geom != 'n_caltech' else 10 * N
    evs = [make_events(n_ev, shape, seed=100 + i) for i in range(B)]
    ranges, off = [], 0
    for e in evs:
        i0, i1 = vis.chunk_bounds(e.shape[0], N)
        ranges += [(off + a, off + b) for a, b in zip(i0, i1)]
        off += e.shape[0]
    ev_d = torch.from_numpy(np.concatenate(evs)).cuda()
    rng = torch.tensor(ranges, dtype=torch.int64).cuda()
    kw = dict(grayscale=False, count_non_zero=g['count_non_zero'],
              background_mask=g['background_mask'])
    frames, raw, kept, stats = vis.events_to_frames_device(ev_d, rng, shape, return_counts=True,
                                                           return_stats=True, max_frame_events=N,
                                                           **kw)
    torch.cuda.synchronize()
    lens = np.array([b - a for a, b in ranges])
    np.testing.assert_array_equal(stats['sum'], lens)            # every event lands in one bin
    np.testing.assert_array_equal(raw.sum(dim=(1, 2, 3)).cpu().numpy(), lens)
    assert (kept <= raw).all()
    for s in (0, B - 1):
        want = oe.events2frames(evs[s], 'event_count', 'event_histogram', shape=shape, N=N, **kw)
        f0 = sum(len(vis.chunk_bounds(e.shape[0], N)[0]) for e in evs[:s])
        np.testing.assert_array_equal(frames[f0:f0 + want.shape[0]].cpu().numpy(), want)


def test_center_events_matches_reference_fixture(hip):
    from conftest import GOLDEN
    from eventclip_amd import vis
    z = np.load(os.path.join(GOLDEN, 'event_utils.npz'))
    res = tuple(int(v) for v in z['resolution'])
    for i in range(int(z['n_cases'])):
        got = vis.center_events(z[f'in{i}'].copy(), res)
        np.testing.assert_array_equal(got, z[f'center{i}'])     # float32, bit for bit


def test_tta_views_match_flipped_oracle(hip):
    """h-/t-flip views (event2img.py:94-112) as kernel flags + reversed chunk ranges against
    events2frames of the reference-style flipped copies."""
    import torch
    from eventclip_amd.event2img import Event2ImagePipeline
    from eventclip_amd.synthetic import make_batch
    from oracle import event_utils as eu
    from oracle import events as oe
    res = (36, 52)
    qa = dict(max_imgs=4, N=900, split_method='event_count', convert_method='event_histogram',
              grayscale=False, count_non_zero=False, background_mask=True)
    evs = make_batch(3, [2300, 700, 3151], res, seed=77)
    pipe = Event2ImagePipeline(res, 3600, qa, n_px=224)
    ev_d = torch.from_numpy(np.concatenate(evs)).cuda()
    n_events = [e.shape[0] for e in evs]
    combos = ((False, False), (True, False), (False, True), (True, True))
    for (h, t) in combos:
        fr, ri, vm = pipe.plan(n_events, tflip=t)
        frames = pipe.frames(ev_d, fr.cuda(), hflip=h, tflip=t).cpu().numpy()
        want = []
        for ev in evs:
            v = ev.copy()
            if h:
                v = eu.hflip_events(v, res)
            if t:
                v = eu.tflip_events(v)
            want.append(oe.events2frames(v, 'event_count', 'event_histogram', shape=res, N=900,
                                         grayscale=False, count_non_zero=False,
                                         background_mask=True))
        np.testing.assert_array_equal(frames, np.concatenate(want))
    views = pipe.tta(ev_d, n_events)
    assert len(views) == 4 and all(tuple(v['img'].shape) == (3, 4, 3, 224, 224) for v in views)


def test_pipeline_center_flag(hip):
    import torch
    from eventclip_amd.event2img import Event2ImagePipeline
    from eventclip_amd.synthetic import make_batch
    from oracle import event_utils as eu
    from oracle import events as oe
    res = (36, 52)
    qa = dict(max_imgs=2, N=900, split_method='event_count', convert_method='event_histogram',
              grayscale=True, count_non_zero=False, background_mask=True)
    evs = make_batch(2, [1000, 1900], res, seed=5)
    for e in evs:
        e[:, 0] = 2 + e[:, 0] % 30       # off-centre window
        e[:, 1] = 1 + e[:, 1] % 20
    pipe = Event2ImagePipeline(res, 1800, qa, n_px=224)
    ev_d = torch.from_numpy(np.concatenate(evs)).cuda()
    fr, ri, vm = pipe.plan([1000, 1900])
    from eventclip_amd import vis
    offs = torch.tensor([[0, 1000], [1000, 2900]], dtype=torch.int64).cuda()
    vis.center_events_device(ev_d, offs, res)
    got = pipe.frames(ev_d, fr.cuda()).cpu().numpy()
    want = np.concatenate([oe.events2frames(eu.center_events(e.copy(), res), 'event_count',
                                            'event_histogram', shape=res, N=900, grayscale=True,
                                            count_non_zero=False, background_mask=True)
                           for e in evs])
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize('packed', [False, True])
@pytest.mark.parametrize('shape,n,frames', [((480, 640), 70000, 5), ((480, 640), 41234, 3), ((300, 640), 90001, 2)])
def test_long_frames_band_sorted_path(shape, n, frames, packed, hip):
    """Frames too long for the LDS event cache: the band-sorted scratch path, the streaming path
    (no workspace) and the oracle all agree bit for bit, including TTA flags, polarity-0 events and
    more frames than resident workgroups' slots are ever reused for."""
    import torch
    from eventclip_amd import vis
    from eventclip_amd.synthetic import make_events
    from oracle import events as oe
    ev = np.concatenate([make_events(n, shape, seed=40 + i) for i in range(frames)])
    ev[::97, 3] = 0                                                  # counted in neither channel
    rng = torch.tensor([[i * n, (i + 1) * n] for i in range(frames)], dtype=torch.int64).cuda()
    e = torch.from_numpy(vis.pack_events(ev).view(np.int64) if packed else ev).cuda()
    for kw in (dict(), dict(flip_x=True, negate_p=True), dict(count_non_zero=True, background_mask=False)):
        a = vis.events_to_frames_device(e, rng, shape, grayscale=False, return_counts=True, return_stats=True,
                                        max_frame_events=n, **kw)
        b = vis.events_to_frames_device(e, rng, shape, grayscale=False, return_counts=True, return_stats=True,
                                        max_frame_events=n, sort_workspace=False, **kw)
        for x, y in zip(a[:3], b[:3]):
            assert torch.equal(x, y)
        for k in ('sum', 'sumsq', 'nnz', 'max_kept', 'dropped'):
            np.testing.assert_array_equal(a[3][k], b[3][k])
    want = oe.events2frames(ev[:n], 'event_count', 'event_histogram', shape=shape, N=n, grayscale=False)
    got = vis.events_to_frames_device(e, rng, shape, grayscale=False, max_frame_events=n)
    np.testing.assert_array_equal(got[0].cpu().numpy(), want[0])


@pytest.mark.parametrize('hot', [0, 700, 1022, 1023, 1500])
def test_long_frames_skip_their_second_pass_through_the_count_of_counts(hot, hip):
    """Band-sorted frames take the largest surviving count (and the ambiguous-count tally) from a histogram of
    counts built in pass 1 instead of re-binning every band a second time; a frame with a count of 1023 or more
    runs the real pass.  Frames and statistics equal the streaming path's (which always runs it) and the oracle's,
    with hot pixels on both sides of the histogram's last slot and of the hot-pixel threshold."""
    import torch
    from eventclip_amd import vis
    from eventclip_amd.synthetic import make_events
    from oracle import events as oe
    shape, n, frames = (480, 640), 70000, 3
    ev = np.concatenate([make_events(n, shape, seed=90 + i) for i in range(frames)])
    if hot:
        for i in range(frames):                       # `hot` events of one polarity on one pixel, a second pixel half as hot
            blk = ev[i * n:(i + 1) * n]
            blk[:hot, 0], blk[:hot, 1], blk[:hot, 3] = 321, 123 + i, 1
            blk[hot:hot + hot // 2, 0], blk[hot:hot + hot // 2, 1], blk[hot:hot + hot // 2, 3] = 17, 400, -1
    rng = torch.tensor([[i * n, (i + 1) * n] for i in range(frames)], dtype=torch.int64).cuda()
    e = torch.from_numpy(ev).cuda()
    for kw in (dict(), dict(count_non_zero=True), dict(thresh=3.0), dict(thresh=0.0)):
        a = vis.events_to_frames_device(e, rng, shape, grayscale=False, return_stats=True, max_frame_events=n, **kw)
        b = vis.events_to_frames_device(e, rng, shape, grayscale=False, return_stats=True, max_frame_events=n,
                                        sort_workspace=False, **kw)
        assert torch.equal(a[0], b[0])
        for k in ('sum', 'sumsq', 'nnz', 'max_kept', 'dropped', 'ambiguous'):
            np.testing.assert_array_equal(a[1][k], b[1][k], err_msg=k)
        want = oe.events2frames(ev[:n], 'event_count', 'event_histogram', shape=shape, N=n, grayscale=False,
                                **{k: v for k, v in kw.items()})
        np.testing.assert_array_equal(a[0][0].cpu().numpy(), want[0])


def test_band_sorted_many_frames_reuse_slots(hip):
    """More frames than CUs: every workgroup walks several frames through its one scratch slot."""
    import torch
    from eventclip_amd import vis
    from eventclip_amd.synthetic import make_events
    shape, n, frames = (480, 640), 30000, 600
    base = [make_events(n, shape, seed=70 + i) for i in range(7)]
    ev = np.concatenate([base[i % 7] for i in range(frames)])
    rng = torch.tensor([[i * n, (i + 1) * n] for i in range(frames)], dtype=torch.int64).cuda()
    e = torch.from_numpy(ev).cuda()
    a = vis.events_to_frames_device(e, rng, shape, max_frame_events=n)
    b = vis.events_to_frames_device(e, rng, shape, max_frame_events=n, sort_workspace=False)
    assert torch.equal(a, b)
    assert torch.equal(a[:7], a[7 * 84:7 * 85])


@pytest.mark.parametrize('seed', range(24))
def test_randomised_geometries_against_oracle(seed, hip):
    """Random sensor shapes, chunk sizes, thresholds, colour maps and flags: every path of the kernel
    (single band, LDS event cache, band-sorted scratch, streaming) against the C oracle, bit for bit."""
    import torch
    from eventclip_amd import vis
    from eventclip_amd.synthetic import make_events
    from oracle import events as oe
    rng = np.random.default_rng(9000 + seed)
    H, W = int(rng.integers(3, 500)), int(rng.integers(4, 160)) * 4 if seed % 3 else int(rng.integers(5, 641))
    N = int(rng.integers(50, 60000))
    n_ev = int(N * rng.uniform(0.3, 4.2))
    ev = make_events(n_ev, (H, W), seed=seed, hot_pixels=int(rng.integers(0, 6)), hot_frac=float(rng.uniform(0.001, 0.05)),
                     p_zero_frac=float(rng.choice([0., 0.02])))
    gray = [True, False, 200, [40, 180, 90]][int(rng.integers(0, 4))]
    kw = dict(N=N, grayscale=gray, count_non_zero=bool(rng.integers(0, 2)), background_mask=bool(rng.integers(0, 2)),
              thresh=float(rng.choice([10., 3., 0.])))
    want, wraw, wkept = oe.events2frames(ev, 'event_count', 'event_histogram', shape=(H, W), return_counts=True, **kw)
    idx0, idx1 = vis.chunk_bounds(n_ev, N)
    r = torch.tensor(np.stack([idx0, idx1], 1), dtype=torch.int64).cuda()
    e = torch.from_numpy(ev).cuda()
    nmax = max(b - a for a, b in zip(idx0, idx1))
    for mfe, ws in ((0, False), (nmax, True), (nmax, False)):
        got, raw, kept, stats = vis.events_to_frames_device(
            e, r, (H, W), grayscale=gray, thresh=kw['thresh'], count_non_zero=kw['count_non_zero'],
            background_mask=kw['background_mask'], return_counts=True, return_stats=True, max_frame_events=mfe,
            sort_workspace=ws)
        np.testing.assert_array_equal(raw.cpu().numpy(), wraw)
        if int(stats['ambiguous'].sum()) == 0:       # a count exactly at the float64 threshold: order of summation decides
            np.testing.assert_array_equal(kept.cpu().numpy(), wkept)
            np.testing.assert_array_equal(got.cpu().numpy(), want)
    # the frames-only product call (no debug outputs): the 10-bit kernels -- whole frame or row bands -- where the
    # geometry allows them, statistics from the binning; same frames, same statistics as the debug path above
    got2, stats2 = vis.events_to_frames_device(
        e, r, (H, W), grayscale=gray, thresh=kw['thresh'], count_non_zero=kw['count_non_zero'],
        background_mask=kw['background_mask'], return_stats=True, max_frame_events=nmax, sort_workspace=True)
    assert torch.equal(got2, got)
    for k in ('sum', 'sumsq', 'nnz', 'max_kept', 'dropped', 'ambiguous'):
        np.testing.assert_array_equal(stats2[k], stats[k], err_msg=k)


@pytest.mark.gpu
@pytest.mark.parametrize('shape,hint', [((480, 640), 70000), ((180, 240), 20000), ((100, 120), 12500)])
def test_ragged_batches_through_the_persistent_kernels(shape, hint, hip):
    """The persistent 10-bit kernels (whole frame / row bands) walk several frames per workgroup with the next
    frame's events in flight: empty frames, one-event frames, frames longer than the caller's max_frame_events
    hint (left to the 32-bit kernel) and ordinary ones in one launch, more frames than workgroups, against the
    streaming path (no workspace: the round-1 kernel) and, frame by frame, the C oracle."""
    import torch
    from eventclip_amd import vis
    from eventclip_amd.synthetic import make_events
    from oracle import events as oe
    rng = np.random.default_rng(5)
    lengths = [0, 1, 5, hint, hint + hint // 3, 300, hint - 1, 0, 4097, 8192, 8193, hint // 2] * 30
    rng.shuffle(lengths)
    evs = [make_events(max(n, 1), shape, seed=300 + i, hot_pixels=i % 3, p_zero_frac=0.01 * (i % 2))[:n] for i, n in enumerate(lengths)]
    # a pixel past the 10-bit fields' 1023 in frames of both rounds of the workgroups' walks (256 CUs): flagged, redone
    # by the 32-bit kernel
    long_enough = [i for i, n in enumerate(lengths) if 4097 <= n <= hint]
    overflowing = [i for i in long_enough if i < 256][:2] + [i for i in long_enough if i >= 256][:3]
    for i in overflowing:
        evs[i][:1100, 0], evs[i][:1100, 1], evs[i][:1100, 3] = 7, 9, 1
    assert any(i >= 256 for i in overflowing)
    ev = np.concatenate(evs)
    ends = np.cumsum(lengths)
    r = torch.tensor(np.stack([ends - np.array(lengths), ends], 1), dtype=torch.int64).cuda()
    e = torch.from_numpy(ev).cuda()
    for packed in (False, True):
        src = torch.from_numpy(vis.pack_events(ev).view(np.int64)).cuda() if packed else e
        a, sa = vis.events_to_frames_device(src, r, shape, grayscale=False, return_stats=True, max_frame_events=hint)
        b, sb = vis.events_to_frames_device(src, r, shape, grayscale=False, return_stats=True, max_frame_events=hint,
                                            sort_workspace=False)
        assert torch.equal(a, b)
        for k in ('sum', 'sumsq', 'nnz', 'max_kept', 'dropped', 'ambiguous'):
            np.testing.assert_array_equal(sa[k], sb[k], err_msg=k)
    got = a.cpu().numpy()
    for i in [0, 3, 7, 11, 100, len(lengths) - 1] + overflowing:
        if lengths[i] == 0:
            continue
        want = oe.events2frames(evs[i], 'event_count', 'event_histogram', shape=shape, N=max(lengths[i], 1), grayscale=False)
        np.testing.assert_array_equal(got[i], want[0])
