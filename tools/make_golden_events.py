"""Generate tests/golden/events_*.npz by running the REFERENCE's own
datasets/vis.py (imported from /root/reference, build container only).

Each fixture holds the inputs (events, compactly) and what the reference
returned for them; big outputs are additionally summarised by sha256 so the
files stay small.  Nothing of the reference's source text is stored.

    python tools/make_golden_events.py
"""
import hashlib
import importlib.util
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from eventclip_amd.synthetic import make_events  # noqa: E402

warnings.filterwarnings('ignore')
spec = importlib.util.spec_from_file_location('refvis', '/root/reference/datasets/vis.py')
vis = importlib.util.module_from_spec(spec)
spec.loader.exec_module(vis)


# ---- the float32 stage of the reference's pinned numpy 1.25 (environment.yml:49) ----
# Under numpy < 2 `hist.astype(np.float32) / hist.max()` (vis.py:27) stays float32 (value-based
# casting: an int64 SCALAR does not upcast a float32 array) and so does everything after it; under
# numpy >= 2 (this image) the np.int64 scalar promotes the stage to float64.  numpy 2.2 has no
# legacy-promotion switch, so the SAME reference code is run a second time on a module copy whose
# `np.stack` hands back integer arrays with a `.max()` that returns a python int -- a weak scalar,
# which NEP 50 treats exactly as numpy 1.x treated the int64 scalar.  Nothing else differs.
class _WeakMax(np.ndarray):
    def max(self, *a, **k):
        r = np.ndarray.max(self.view(np.ndarray), *a, **k)
        return int(r) if np.ndim(r) == 0 else r


class _LegacyNumpy:
    def __getattr__(self, k):
        return getattr(np, k)

    @staticmethod
    def stack(arrays, axis=0, **kw):
        out = np.stack(arrays, axis=axis, **kw)
        return out.view(_WeakMax) if out.dtype.kind == 'i' else out


spec32 = importlib.util.spec_from_file_location('refvis_f32', '/root/reference/datasets/vis.py')
vis32 = importlib.util.module_from_spec(spec32)
spec32.loader.exec_module(vis32)
vis32.np = _LegacyNumpy()


def ref_counts(ev, N, shape):
    """Raw per-chunk counts through the reference's own parse/split + bincount lines."""
    x, y, t, p = vis.parse_events(ev)
    idx0, idx1, _, _ = vis.split_event_count(t, N)
    H, W = shape
    out = []
    for i0, i1 in zip(idx0, idx1):
        xx, yy, pp = x[i0:i1], y[i0:i1], p[i0:i1]
        pos = np.bincount(xx[pp > 0] + yy[pp > 0] * W, minlength=H * W).reshape(H, W)
        neg = np.bincount(xx[pp < 0] + yy[pp < 0] * W, minlength=H * W).reshape(H, W)
        out.append(np.stack([pos, neg], -1))
    return np.stack(out).astype(np.int32)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def case(name, ev, shape, N, grayscale=True, count_non_zero=False, background_mask=True,
         store_full=True):
    kw = dict(N=N, grayscale=grayscale, count_non_zero=count_non_zero,
              background_mask=background_mask)
    frames = vis.events2frames(ev.copy(), 'event_count', 'event_histogram', shape=shape,
                               max_imgs=10, **dict(kw))
    frames32 = np.asarray(vis32.events2frames(ev.copy(), 'event_count', 'event_histogram', shape=shape,
                                              max_imgs=10, **dict(kw)))
    assert frames32.dtype == np.uint8 and frames32.shape == frames.shape
    raw = ref_counts(ev.copy(), N, shape)
    d = dict(
        name=name, shape=np.array(shape), N=N,
        grayscale=np.array(grayscale if not isinstance(grayscale, bool) else int(grayscale)),
        grayscale_is_bool=isinstance(grayscale, bool), count_non_zero=count_non_zero,
        background_mask=background_mask,
        ev_x=ev[:, 0].copy(), ev_y=ev[:, 1].copy(), ev_t=ev[:, 2].copy(), ev_p=ev[:, 3].copy(),
        frames_sha256=sha(frames), raw_sha256=sha(raw), n_frames=frames.shape[0],
        numpy_version=np.__version__, float_stage='float64',
        frames_f32_sha256=sha(frames32),
        f32_differs=int((frames32 != frames).sum()),
    )
    # integer-valued coordinates compress to int16
    if np.all(ev[:, 0] == np.floor(ev[:, 0])) and np.all(ev[:, 1] == np.floor(ev[:, 1])):
        d['ev_x'] = ev[:, 0].astype(np.int16)
        d['ev_y'] = ev[:, 1].astype(np.int16)
        d['ev_p'] = ev[:, 3].astype(np.int8)
        d['ev_t'] = np.array([ev[0, 2], ev[-1, 2]], dtype=np.float32)  # t is unused downstream
    if store_full:
        d['frames'] = frames
        d['raw'] = raw
    if d['f32_differs']:
        idx = np.flatnonzero(frames32.ravel() != frames.ravel())
        d['f32_diff_index'] = idx.astype(np.int64)          # where the float32 stage rounds differently
        d['f32_diff_value'] = frames32.ravel()[idx]
    return d


def main():
    out_dir = os.path.join(ROOT, 'tests', 'golden')
    os.makedirs(out_dir, exist_ok=True)
    cases = []
    cal, car, nin, tiny = (180, 240), (100, 120), (480, 640), (36, 52)

    # dataset configs of the reference (configs/zsclip/*.py), plus RGB polarity
    cases.append(case('caltech_gray_rem_half_dropped', make_events(50000, cal, 1), cal, 20000))
    cases.append(case('caltech_gray_rem_half_plus1', make_events(50001, cal, 2), cal, 20000))
    cases.append(case('caltech_rgb', make_events(45000, cal, 3), cal, 20000, grayscale=False,
                      store_full=False))
    cases.append(case('cars_short', make_events(12500, car, 4), car, 30000,
                      count_non_zero=True, background_mask=False))
    cases.append(case('cars_rgb_bg', make_events(61000, car, 5), car, 30000, grayscale=False,
                      count_non_zero=True, background_mask=True))
    cases.append(case('nin_gray', make_events(140000, nin, 6), nin, 70000, store_full=False))
    cases.append(case('nin_rgb_nobg_cnz', make_events(110000, nin, 7), nin, 70000,
                      grayscale=False, count_non_zero=True, background_mask=False,
                      store_full=False))
    # all flag combinations on a small sensor with small maxima (many .5 ties)
    k = 10
    for gray in (True, False, 200, [90, 127, 255]):
        for cnz in (False, True):
            for bg in (False, True):
                k += 1
                tag = 'g' + ''.join(ch if ch.isalnum() else '_' for ch in str(gray).replace(' ', '')).strip('_')
                cases.append(case(f'tiny_{tag}_cnz{int(cnz)}_bg{int(bg)}',
                                  make_events(2300, tiny, k, p_zero_frac=0.02), tiny, 900,
                                  grayscale=gray, count_non_zero=cnz, background_mask=bg))
    # edge cases
    ev = make_events(3000, tiny, 40, hot_pixels=0)
    ev[:, 0], ev[:, 1] = 7, 11                       # every event on one pixel -> removed -> NaN path
    cases.append(case('one_pixel_all_events', ev, tiny, 1000))
    ev = make_events(1500, tiny, 41, hot_pixels=0)
    ev[:, 3] = 0.                                    # polarity 0: counted in neither channel
    cases.append(case('polarity_zero_only', ev, tiny, 1000))
    H, W = tiny
    xs, ys = np.meshgrid(np.arange(W), np.arange(H))
    ev = np.stack([xs.ravel(), ys.ravel(), np.linspace(0, 0.1, H * W), np.ones(H * W)], 1)
    ev = np.concatenate([ev, ev * [1, 1, 1, -1]], 0).astype(np.float32)
    cases.append(case('constant_image', ev, tiny, 2 * H * W + 5))   # std == 0, one chunk
    ev = make_events(4000, tiny, 42, hot_pixels=1, hot_frac=0.2)
    cases.append(case('one_hot_pixel', ev, tiny, 4000, count_non_zero=True))
    ev = make_events(2000, tiny, 43)
    ev[:, 0] += np.random.default_rng(0).uniform(0, 0.99, 2000).astype(np.float32)
    ev[:, 1] += 0.5                                  # fractional coordinates truncate (vis.py:50)
    cases.append(case('fractional_coords', ev, tiny, 800))
    cases.append(case('exactly_N', make_events(900, tiny, 44), tiny, 900))
    cases.append(case('fewer_than_N', make_events(899, tiny, 45), tiny, 900))
    cases.append(case('rem_exactly_half', make_events(900 * 3 + 450, tiny, 46), tiny, 900))
    cases.append(case('rem_half_plus1', make_events(900 * 3 + 451, tiny, 47), tiny, 900))

    # every (positive, negative) count pair up to a maximum of 6 / 10 / 12: the .5 ties where the
    # float32 stage (numpy 1.25) and the float64 stage (numpy >= 2) round differently
    for mx, gray, bg in ((6, True, False), (12, True, False), (6, False, True), (10, False, True),
                         (12, False, True)):
        rows = []
        pix = 0
        for c0 in range(mx + 1):
            for c1 in range(mx + 1):
                x, y = pix % tiny[1], pix // tiny[1]
                rows += [[x, y, 0., 1.]] * c0 + [[x, y, 0., -1.]] * c1
                pix += 1
        ev = np.array(rows, dtype=np.float32)
        ev = ev[np.random.default_rng(mx).permutation(len(ev))]
        ev[:, 2] = np.linspace(0, 0.1, len(ev))
        cases.append(case(f'ties_max{mx}_{"gray" if gray else "rgb"}_bg{int(bg)}', ev, tiny, len(ev) + 7,
                          grayscale=gray, background_mask=bg))

    for i, d in enumerate(cases):
        np.savez_compressed(os.path.join(out_dir, f'events_{i:02d}_{d["name"]}.npz'), **d)
    tot = sum(os.path.getsize(os.path.join(out_dir, f)) for f in os.listdir(out_dir)
              if f.startswith('events_'))
    print(f'wrote {len(cases)} fixtures, {tot / 1024:.0f} KiB; float32 stage differs from float64 in '
          f'{sum(int(d["f32_differs"]) for d in cases)} bytes over {sum(1 for d in cases if d["f32_differs"])} fixtures')


if __name__ == '__main__':
    main()
