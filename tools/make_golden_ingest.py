"""tests/golden/ingest.npz: N-ImageNet-style structured event records and what the reference's
datasets/imagenet.py:load_event returns for them (run here, where /root/reference exists).

    python tools/make_golden_ingest.py
"""
import importlib.util
import os
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# datasets/imagenet.py only needs its sibling for the class it subclasses; load_event is numpy only
pkg = types.ModuleType('datasets')
pkg.__path__ = ['/root/reference/datasets']
sys.modules['datasets'] = pkg
cal = types.ModuleType('datasets.caltech')
cal.NCaltech101 = type('NCaltech101', (), {})
sys.modules['datasets.caltech'] = cal
spec = importlib.util.spec_from_file_location('datasets.imagenet', '/root/reference/datasets/imagenet.py')
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

rng = np.random.default_rng(77)
out = {}
cases = [('u01', np.uint8, (0, 2)),      # polarity 0 / 1 as in the released files
         ('bool', np.bool_, (0, 2)),
         ('i8pm', np.int8, (-1, 2)),     # -1 / 0 / 1 stored signed: the uint8 cast of imagenet.py:15 applies
         ('ones', np.uint8, (1, 2))]     # all-positive sample
for i, (name, pdt, (lo, hi)) in enumerate(cases):
    n = 400 + 37 * i
    dt = np.dtype([('x', np.uint16), ('y', np.uint16), ('t', np.int64), ('p', pdt)])
    rec = np.zeros(n, dtype=dt)
    rec['x'] = rng.integers(0, 640, n)
    rec['y'] = rng.integers(0, 480, n)
    rec['t'] = np.sort(rng.integers(1_000, 60_000, n))
    rec['p'] = rng.integers(lo, hi, n).astype(pdt)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, 'ev.npz')
        np.savez(path, event_data=rec)
        exp = ref.load_event(path)
    out[f'rec{i}'] = rec
    out[f'exp{i}'] = exp
out['n_cases'] = np.array(len(cases))
dst = os.path.join(ROOT, 'tests', 'golden', 'ingest.npz')
np.savez_compressed(dst, **out)
print('wrote', dst, os.path.getsize(dst) // 1024, 'KiB')
