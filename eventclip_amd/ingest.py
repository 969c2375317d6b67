"""Event file ingest: the step right before the hot path at test time.

Mirrors the reference readers:
  * N-Caltech101 / N-Cars: ``np.load(path).astype(np.float32)``, a [n, 4] (x, y, t, p) array
    (datasets/caltech.py:148-151);
  * N-ImageNet: ``np.load(path)['event_data']``, a structured array with integer x, y, t
    (microseconds) and p in {0, 1}; stacked to float [n, 4], t / 1e6, polarity 0 -> -1 unless
    negative polarities are present (datasets/imagenet.py:8-27).
``packed=True`` returns the 8-byte form of include/eventclip_hip.h instead (half the bytes over
PCIe and through the binning kernel); for N-ImageNet it is built from the integer fields
directly, without the float64 detour.
"""
import numpy as np

from . import vis


def load_npy_events(path, packed=False):
    """caltech.py:148-151."""
    ev = np.load(path).astype(np.float32)
    return vis.pack_events(ev) if packed else ev


def load_npz_events(path, packed=False, key='event_data'):
    """imagenet.py:8-27 (float64 [n, 4] like the reference unless packed)."""
    rec = np.load(path)[key]
    if packed:
        return vis.pack_structured(rec['x'], rec['y'], rec['t'], rec['p'])
    event = np.stack([rec['x'], rec['y'], rec['t'], rec['p'].astype(np.uint8)], 1)   # :11-16
    event = event.astype(float)                                                       # :18
    event[:, 2] /= 1e6                                                                # :21
    if event[:, 3].min() >= -0.5:                                                     # :24-25
        event[:, 3][event[:, 3] <= 0.5] = -1
    return event


def load_events(path, packed=False):
    """Dispatch on the extension: .npy (N-Caltech101 / N-Cars) or .npz (N-ImageNet)."""
    if str(path).endswith('.npz'):
        return load_npz_events(path, packed)
    return load_npy_events(path, packed)
