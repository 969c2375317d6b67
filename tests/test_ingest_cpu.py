"""Event ingest: file readers and the packed 8-byte event form, host side."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, event_fixture_paths, load_event_fixture
from eventclip_amd import ingest, vis
from oracle import event_utils as eu


def cases():
    z = np.load(os.path.join(GOLDEN, 'ingest.npz'))
    return [(z[f'rec{i}'], z[f'exp{i}']) for i in range(int(z['n_cases']))]


@pytest.mark.parametrize('i', range(4))
def test_npz_reader_matches_reference(i, tmp_path):
    rec, exp = cases()[i]
    np.testing.assert_array_equal(eu.load_event_npz(rec), exp)          # oracle vs reference
    path = str(tmp_path / 'ev.npz')
    np.savez(path, event_data=rec)
    got = ingest.load_events(path)
    assert got.dtype == exp.dtype
    np.testing.assert_array_equal(got, exp)                              # product reader vs reference


@pytest.mark.parametrize('i', range(4))
def test_packed_from_records_equals_packed_from_floats(i, tmp_path):
    rec, exp = cases()[i]
    path = str(tmp_path / 'ev.npz')
    np.savez(path, event_data=rec)
    packed = ingest.load_events(path, packed=True)
    assert packed.dtype == np.uint64 and packed.shape == (len(rec),)
    np.testing.assert_array_equal(packed, vis.pack_events(exp))
    x, y, code, t_us = eu.packed_fields(exp)
    np.testing.assert_array_equal(packed & np.uint64(0xffff), x.astype(np.uint64))
    np.testing.assert_array_equal((packed >> np.uint64(16)) & np.uint64(0xffff), y.astype(np.uint64))
    np.testing.assert_array_equal((packed >> np.uint64(32)) & np.uint64(3), code.astype(np.uint64))
    np.testing.assert_array_equal(packed >> np.uint64(34), t_us.astype(np.uint64))
    back = vis.unpack_events(packed)
    np.testing.assert_array_equal(back[:, :2], exp[:, :2].astype(np.float32))
    # only the sign of p reaches the histogram (vis.py:10,12); int8 -1 becomes 255 at imagenet.py:15
    np.testing.assert_array_equal(np.sign(back[:, 3]), np.sign(exp[:, 3]))
    np.testing.assert_allclose(back[:, 2], exp[:, 2], rtol=0, atol=1e-6)


def test_npy_reader(tmp_path):
    ev = load_event_fixture(event_fixture_paths()[0])[0]
    path = str(tmp_path / 'ev.npy')
    np.save(path, ev.astype(np.float64))
    got = ingest.load_events(path)
    assert got.dtype == np.float32                                       # caltech.py:151
    np.testing.assert_array_equal(got, ev.astype(np.float32))


def test_pack_roundtrip_on_event_fixtures():
    for path in event_fixture_paths()[:8]:
        ev = load_event_fixture(path)[0]
        inside = (ev[:, 0] >= 0) & (ev[:, 1] >= 0)
        ev = ev[inside]
        x, y, code, _ = eu.packed_fields(ev)
        packed = vis.pack_events(ev)
        np.testing.assert_array_equal((packed >> np.uint64(32)) & np.uint64(3), code.astype(np.uint64))
        back = vis.unpack_events(packed)
        np.testing.assert_array_equal(back[:, 0], x)
        np.testing.assert_array_equal(back[:, 1], y)
        np.testing.assert_array_equal(np.sign(back[:, 3]), np.sign(ev[:, 3].astype(np.int32)))


def test_pack_rejects_what_it_cannot_represent():
    with pytest.raises(ValueError):
        vis.pack_events(np.array([[1.5, 2, 0, 1]], dtype=np.float32))
    with pytest.raises(ValueError):
        vis.pack_events(np.array([[-1, 2, 0, 1]], dtype=np.float32))
    with pytest.raises(ValueError):
        vis.pack_structured([70000], [1], [0], [1])
