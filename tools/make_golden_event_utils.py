"""tests/golden/event_utils.npz from the reference's own datasets/utils.py (numpy only).

    python tools/make_golden_event_utils.py
"""
import copy
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from eventclip_amd.synthetic import make_events  # noqa: E402

spec = importlib.util.spec_from_file_location('refutils', '/root/reference/datasets/utils.py')
ru = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ru)

out = {}
res = (36, 52)
for i, (n, lo, hi) in enumerate([(700, (3, 2), (40, 30)), (500, (10, 0), (51, 35)), (300, (0, 5), (20, 20))]):
    ev = make_events(n, res, seed=50 + i, hot_pixels=0)
    # squeeze the events into a sub-window so centering has something to do
    ev[:, 0] = lo[0] + ev[:, 0] % (hi[0] - lo[0] + 1)
    ev[:, 1] = lo[1] + ev[:, 1] % (hi[1] - lo[1] + 1)
    ev[:, 2] += 0.25
    out[f'in{i}'] = ev.copy()
    out[f'center{i}'] = ru.center_events(ev.copy(), resolution=res)
    out[f'hflip{i}'] = ru.random_flip_events_along_x(ev.copy(), resolution=res, p=1.)
    out[f'tflip{i}'] = ru.random_time_flip_events(ev.copy(), p=1.)
    h = ru.random_flip_events_along_x(ev.copy(), resolution=res, p=1.)
    out[f'htflip{i}'] = ru.random_time_flip_events(copy.deepcopy(h), p=1.)
    # NCaltech101._augment_events (caltech.py:153-163): the reference's own functions in its order, with
    # the global numpy stream seeded so the draws can be replayed
    for seed in (0, 1, 2, 3):
        for ft in (False, True):
            np.random.seed(1000 * i + 10 * seed + int(ft))
            a = ev.copy()
            if ft:
                a = ru.random_time_flip_events(a)
            a = ru.random_shift_events(a, max_shift=12, resolution=res)
            a = ru.random_flip_events_along_x(a, resolution=res)
            out[f'aug{i}_{seed}_{int(ft)}'] = a
out['aug_max_shift'] = np.array(12)
out['resolution'] = np.array(res)
out['n_cases'] = np.array(3)
np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'event_utils.npz'), **out)
print('wrote event_utils.npz', os.path.getsize(os.path.join(ROOT, 'tests', 'golden', 'event_utils.npz')) // 1024, 'KiB')
