import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_event_fixture(path):
    """-> (events float32 [n,4], kwargs for events2frames, dict of expected outputs)."""
    z = np.load(path, allow_pickle=False)
    x = z['ev_x'].astype(np.float32)
    y = z['ev_y'].astype(np.float32)
    p = z['ev_p'].astype(np.float32)
    n = x.shape[0]
    if z['ev_t'].shape[0] == n:
        t = z['ev_t'].astype(np.float32)
    else:  # t only orders the events; the path never reads its values
        t = np.linspace(float(z['ev_t'][0]), float(z['ev_t'][1]), n).astype(np.float32)
    ev = np.stack([x, y, t, p], 1)
    if bool(z['grayscale_is_bool']):
        gray = bool(z['grayscale'])
    else:
        g = z['grayscale']
        gray = g.tolist() if g.ndim else int(g)
    kw = dict(N=int(z['N']), grayscale=gray, count_non_zero=bool(z['count_non_zero']),
              background_mask=bool(z['background_mask']))
    shape = tuple(int(v) for v in z['shape'])
    exp = dict(frames_sha256=str(z['frames_sha256']), raw_sha256=str(z['raw_sha256']),
               n_frames=int(z['n_frames']))
    if 'frames' in z.files:
        exp['frames'] = z['frames']
        exp['raw'] = z['raw']
    # the reference's float32 stage (numpy 1.x casting; tools/make_golden_events.py)
    exp['frames_f32_sha256'] = str(z['frames_f32_sha256'])
    exp['f32_differs'] = int(z['f32_differs'])
    if exp['f32_differs']:
        exp['f32_diff_index'], exp['f32_diff_value'] = z['f32_diff_index'], z['f32_diff_value']
    return ev, shape, kw, exp


def event_fixture_paths():
    return sorted(glob.glob(os.path.join(GOLDEN, 'events_*.npz')))


@pytest.fixture(scope='session')
def hip():
    """The product's C-ABI library; GPU tests use it, and skip nothing silently."""
    from eventclip_amd import _lib
    return _lib.lib()
