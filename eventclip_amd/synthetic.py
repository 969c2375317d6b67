"""Seeded synthetic inputs for tests and benchmarks (no datasets ship with the repo).

Event streams follow SURVEY.md section 8(d): float32 [n_ev, 4] rows (x, y, t, p) as
the reference's dataset readers produce them (caltech.py:149-151): integer
pixel coordinates, t sorted in seconds, p in {-1, +1}.
"""
import numpy as np

# dataset constants of the reference (caltech.py:52-58, cars.py:30-32, imagenet.py:48-50)
GEOMETRY = {
    'n_caltech': dict(resolution=(180, 240), max_n=225000, max_t=0.325, N=20000,
                      count_non_zero=False, background_mask=True, n_cls=101),
    'n_cars': dict(resolution=(100, 120), max_n=12500, max_t=0.1, N=30000,
                   count_non_zero=True, background_mask=False, n_cls=2),
    'n_imagenet': dict(resolution=(480, 640), max_n=135000, max_t=0.055, N=70000,
                       count_non_zero=False, background_mask=True, n_cls=1000),
}


def make_events(n_ev, resolution, seed=0, max_t=0.3, hot_pixels=4, hot_frac=0.005,
                blob_frac=0.1, p_zero_frac=0.0, blob_sigma=None):
    """One sample's events: 90 % uniform, 10 % in a Gaussian blob (sigma = H/8 unless ``blob_sigma`` pixels),
    plus ``hot_pixels`` pixels that each receive ``hot_frac`` of all events so
    that the hot-pixel removal (vis.py:17-24) has something to remove."""
    H, W = resolution
    rng = np.random.default_rng(seed)
    x = rng.integers(0, W, size=n_ev)
    y = rng.integers(0, H, size=n_ev)
    blob = rng.random(n_ev) < blob_frac
    cx, cy = rng.uniform(0.25, 0.75) * W, rng.uniform(0.25, 0.75) * H
    sig = H / 8. if blob_sigma is None else float(blob_sigma)
    bx = np.clip(np.rint(rng.normal(cx, sig, size=n_ev)), 0, W - 1)
    by = np.clip(np.rint(rng.normal(cy, sig, size=n_ev)), 0, H - 1)
    x = np.where(blob, bx, x)
    y = np.where(blob, by, y)
    if hot_pixels:
        hx = rng.integers(0, W, size=hot_pixels)
        hy = rng.integers(0, H, size=hot_pixels)
        u = rng.random(n_ev)
        for k in range(hot_pixels):
            sel = (u >= k * hot_frac) & (u < (k + 1) * hot_frac)
            x = np.where(sel, hx[k], x)
            y = np.where(sel, hy[k], y)
    p = np.where(rng.random(n_ev) < 0.5, 1., -1.)
    if p_zero_frac > 0:
        p = np.where(rng.random(n_ev) < p_zero_frac, 0., p)
    t = np.sort(rng.uniform(0., max_t, size=n_ev))
    return np.stack([x, y, t, p], axis=1).astype(np.float32)


def make_batch(batch, n_ev, resolution, seed=0, **kw):
    """List of ``batch`` event arrays; ``n_ev`` may be an int or a per-sample list."""
    if np.isscalar(n_ev):
        n_ev = [int(n_ev)] * batch
    return [make_events(int(n), resolution, seed=seed * 100003 + i, **kw)
            for i, n in enumerate(n_ev)]
