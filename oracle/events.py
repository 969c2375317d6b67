"""Oracle for events -> histogram frames (TEST INFRASTRUCTURE, see oracle/__init__.py).

Thin ctypes front end over ``oracle/events_oracle.c``, which restates
``/root/reference/datasets/vis.py`` (make_event_histogram :6-41, parse_events
:44-52, split_event_count :55-72, events2frames :75-117).  The C file is built
on first use with gcc into ``oracle/_build/`` (also by
``__graft_entry__.build()``).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libevents_oracle.so')
_lib = None


def build(force=False):
    src = os.path.join(_HERE, 'events_oracle.c')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-s', '-C', _HERE, '-B', '_build/libevents_oracle.so'])
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.ec_oracle_set_float32_stage.restype = None
        _lib.ec_oracle_set_float32_stage.argtypes = [ctypes.c_int]
        _lib.ec_oracle_np_sum.restype = ctypes.c_double
        _lib.ec_oracle_np_sum.argtypes = [ctypes.c_void_p, ctypes.c_long]
        _lib.ec_oracle_split_event_count.restype = ctypes.c_long
        _lib.ec_oracle_split_event_count.argtypes = [
            ctypes.c_long, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long]
        _lib.ec_oracle_events2frames.restype = ctypes.c_long
        _lib.ec_oracle_events2frames.argtypes = [
            ctypes.c_void_p, ctypes.c_long, ctypes.c_long, ctypes.c_int, ctypes.c_int,
            ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
            ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    return _lib


def np_sum(a):
    """numpy's float64 pairwise add.reduce over a contiguous 1-D run."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    return lib().ec_oracle_np_sum(a.ctypes.data, a.size)


def split_event_count(tot_cnt, N):
    """vis.py:55-72 -> (idx0, idx1) python lists."""
    cap = tot_cnt // max(N, 1) + 2
    i0 = np.zeros(cap, dtype=np.int64)
    i1 = np.zeros(cap, dtype=np.int64)
    f = lib().ec_oracle_split_event_count(tot_cnt, N, i0.ctypes.data, i1.ctypes.data, cap)
    return i0[:f].tolist(), i1[:f].tolist()


def colour_map(grayscale=True):
    """vis.py:95-104 -> (red, blue) uint8[3]."""
    if isinstance(grayscale, np.ndarray) or grayscale:
        v = 127 if isinstance(grayscale, bool) else np.array(grayscale)
        red = np.round(np.ones(3) * v).astype(np.uint8)
        blue = np.round(np.ones(3) * v).astype(np.uint8)
    else:
        red = np.array([255, 0, 0], dtype=np.uint8)
        blue = np.array([0, 0, 255], dtype=np.uint8)
    return red, blue


def events2frames(events, split_method='event_count', convert_method='event_histogram',
                  shape=(180, 240), return_counts=False, **kwargs):
    """Same signature and result as the reference's events2frames (vis.py:75-117).

    ``events`` is float32/float64 [n_ev, 4] (x, y, t, p).  With
    ``return_counts`` also returns the raw and the post-hot-pixel-removal
    counts as int64 [F, H, W, 2].
    """
    grayscale = kwargs.pop('grayscale', True)
    assert split_method == 'event_count'
    if convert_method != 'event_histogram':
        raise NotImplementedError(f'{convert_method} not implemented!')
    N = int(kwargs['N'])
    thresh = float(kwargs.get('thresh', 10.))
    # float_stage='float32' (default): the reference's pinned numpy 1.25 semantics of vis.py:27-39 (value-based
    # casting keeps the stage in float32); 'float64' = numpy >= 2, what importing vis.py in this image gives
    f32 = kwargs.get('float_stage', 'float32') == 'float32'
    cnz = bool(kwargs.get('count_non_zero', False))
    bgm = bool(kwargs.get('background_mask', True))
    red, blue = colour_map(grayscale)
    H, W = shape
    ev = np.ascontiguousarray(events, dtype=np.float32)
    assert ev.ndim == 2 and ev.shape[1] == 4
    n_ev = ev.shape[0]
    if n_ev == 0:
        raise IndexError('empty event array (the reference fails at vis.py:61)')
    max_frames = n_ev // N + 2
    frames = np.zeros((max_frames, H, W, 3), dtype=np.uint8)
    raw = np.zeros((max_frames, H, W, 2), dtype=np.int64) if return_counts else None
    kept = np.zeros((max_frames, H, W, 2), dtype=np.int64) if return_counts else None
    lib().ec_oracle_set_float32_stage(int(f32))
    F = lib().ec_oracle_events2frames(
        ev.ctypes.data, n_ev, N, H, W, thresh, int(cnz), int(bgm), red.ctypes.data,
        blue.ctypes.data, max_frames, frames.ctypes.data,
        raw.ctypes.data if return_counts else None, kept.ctypes.data if return_counts else None)
    lib().ec_oracle_set_float32_stage(0)
    if F == -1:
        raise ValueError('event outside the sensor (the reference fails at vis.py:11)')
    assert F > 0
    if return_counts:
        return frames[:F].copy(), raw[:F].copy(), kept[:F].copy()
    return frames[:F].copy()
