"""The events oracle against the golden vectors produced by the reference's vis.py."""
import hashlib
import os

import numpy as np
import pytest

from conftest import event_fixture_paths, load_event_fixture
from oracle import events as oe


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize('path', event_fixture_paths(), ids=os.path.basename)
def test_oracle_matches_reference_fixture(path):
    ev, shape, kw, exp = load_event_fixture(path)
    # the fixtures' main frames come from the reference's vis.py imported under THIS image's numpy (>= 2:
    # float64 stage); the reference's pinned numpy 1.25 (the default, float32 stage) is the second hash below
    frames, raw, kept = oe.events2frames(ev, 'event_count', 'event_histogram', shape=shape,
                                         return_counts=True, float_stage='float64', **kw)
    assert frames.shape[0] == exp['n_frames']
    assert frames.dtype == np.uint8 and frames.shape[1:] == (*shape, 3)
    assert sha(raw.astype(np.int32)) == exp['raw_sha256']
    assert sha(frames) == exp['frames_sha256']
    if 'frames' in exp:
        np.testing.assert_array_equal(frames, exp['frames'])
        np.testing.assert_array_equal(raw, exp['raw'])
    assert (kept <= raw).all()
    # float32 stage: what the reference computes under its pinned numpy 1.25 (value-based casting)
    f32 = oe.events2frames(ev, 'event_count', 'event_histogram', shape=shape, float_stage='float32', **kw)
    assert sha(f32) == exp['frames_f32_sha256']
    np.testing.assert_array_equal(f32, oe.events2frames(ev, 'event_count', 'event_histogram', shape=shape, **kw))   # = default
    d = np.flatnonzero(f32.ravel() != frames.ravel())
    assert len(d) == exp['f32_differs']
    if len(d):                                   # 1 LSB apart, at exact .5 ties only
        np.testing.assert_array_equal(d, exp['f32_diff_index'])
        np.testing.assert_array_equal(f32.ravel()[d], exp['f32_diff_value'])
        assert np.abs(f32.ravel()[d].astype(int) - frames.ravel()[d].astype(int)).max() == 1


def test_fixture_count():
    paths = event_fixture_paths()
    assert len(paths) >= 35
    # some fixtures must actually separate the two float stages
    assert sum(load_event_fixture(p)[3]['f32_differs'] > 0 for p in paths) >= 3


@pytest.mark.parametrize('tot,N,exp0,exp1', [
    (5, 10, [0], [5]),                       # fewer than N: one chunk (vis.py:60-61)
    (10, 10, [0], [10]),
    (14, 10, [0], [10]),                     # remainder 4 <= N/2 dropped
    (15, 10, [0], [10]),                     # remainder exactly N/2 dropped
    (16, 10, [0, 6], [10, 16]),              # remainder > N/2: overlapping last chunk (vis.py:67-69)
    (30, 10, [0, 10, 20], [10, 20, 30]),
])
def test_split_event_count(tot, N, exp0, exp1):
    i0, i1 = oe.split_event_count(tot, N)
    assert i0 == exp0 and i1 == exp1


def test_out_of_sensor_raises():
    ev = np.array([[5, 5, 0, 1], [300, 5, 0.1, 1]], dtype=np.float32)
    with pytest.raises(ValueError):
        oe.events2frames(ev, 'event_count', 'event_histogram', shape=(36, 52), N=10)


def test_numpy_sum_order():
    # the std() restatement depends on numpy's reduction order staying what it was
    rng = np.random.default_rng(3)
    for n in (5, 100, 129, 8192, 8200, 86400):
        a = rng.random(n) ** 3 * 1e3
        assert oe.np_sum(a) == a.sum()


def test_event_utils_oracle_matches_reference():
    from conftest import GOLDEN
    from oracle import event_utils as eu
    z = np.load(os.path.join(GOLDEN, 'event_utils.npz'))
    res = tuple(int(v) for v in z['resolution'])
    for i in range(int(z['n_cases'])):
        ev = z[f'in{i}']
        np.testing.assert_array_equal(eu.center_events(ev.copy(), res), z[f'center{i}'])
        np.testing.assert_array_equal(eu.hflip_events(ev.copy(), res), z[f'hflip{i}'])
        np.testing.assert_array_equal(eu.tflip_events(ev.copy()), z[f'tflip{i}'])
        views = eu.tta_views(ev.copy(), res)
        np.testing.assert_array_equal(views[3], z[f'htflip{i}'])


def test_event_augmentation_replays_the_reference_stream():
    """Seeded numpy draws (eventclip_amd.augment.draw_event_augment makes the reference's calls in the
    reference's order) + the oracle's restatement == NCaltech101._augment_events' own output."""
    from conftest import GOLDEN
    from eventclip_amd.augment import draw_event_augment
    from oracle import event_utils as eu
    z = np.load(os.path.join(GOLDEN, 'event_utils.npz'))
    res = tuple(int(v) for v in z['resolution'])
    ms = int(z['aug_max_shift'])
    for i in range(int(z['n_cases'])):
        for seed in (0, 1, 2, 3):
            for ft in (0, 1):
                np.random.seed(1000 * i + 10 * seed + ft)
                prm = draw_event_augment(1, ms, bool(ft))[0]
                got = eu.augment_events(z[f'in{i}'], prm, res)
                np.testing.assert_array_equal(got, z[f'aug{i}_{seed}_{ft}'])
