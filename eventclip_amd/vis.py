"""events -> 2-D frames on the MI355X; host-side mirror of the reference's datasets/vis.py.

Same names and argument meaning as the reference:

* ``events2frames(events, split_method, convert_method, shape, **kwargs)``
  (vis.py:75-117) -- numpy in, numpy uint8 [F, H, W, 3] out, computed by the HIP
  kernel behind ``ec_events_to_frames``.
* ``split_event_count(t, N)`` (vis.py:55-72), ``parse_events`` (vis.py:44-52).

plus the batched device entry ``events_to_frames_device`` used by the pipeline.
There is no CPU fallback.
"""
import ctypes

import numpy as np

from . import _lib


def parse_events(events):
    """vis.py:44-52: accept an [n, 4] (x, y, t, p) array or a dict of columns.
    Returns one contiguous float32 [n, 4] array (the integer truncation of
    x, y, p happens in the kernel)."""
    if isinstance(events, dict):
        cols = [np.asarray(events[k]) for k in ('x', 'y', 't', 'p')]
        events = np.stack(cols, axis=1)
    ev = np.ascontiguousarray(events, dtype=np.float32)
    if ev.ndim != 2 or ev.shape[1] != 4:
        raise ValueError(f'events must be [n, 4] (x, y, t, p), got {ev.shape}')
    return ev


def chunk_bounds(tot_cnt, N):
    """Chunk index pairs of vis.py:55-72 from the event count alone."""
    tot_cnt, N = int(tot_cnt), int(N)
    if tot_cnt < N:                       # vis.py:60-61
        return [0], [tot_cnt]
    idx = list(range(0, tot_cnt, N))      # vis.py:64
    idx1, idx0 = idx[1:], idx[:-1]        # vis.py:65
    if tot_cnt - idx[-1] > N * 0.5:       # vis.py:67-69: overlapping last chunk
        idx0.append(tot_cnt - N)
        idx1.append(tot_cnt)
    return idx0, idx1


def split_event_count(t, N=30000):
    """Same return as the reference: (idx0, idx1, t0, t1)."""
    t = np.asarray(t)
    idx0, idx1 = chunk_bounds(len(t), N)
    if len(t) < N:
        return idx0, idx1, [t[0]], [t[-1]]
    return idx0, idx1, t[idx0], t[np.array(idx1) - 1]


def colour_map(grayscale=True):
    """vis.py:95-104 -> (red, blue) uint8[3]."""
    if grayscale:
        v = 127 if isinstance(grayscale, bool) else np.array(grayscale)
        red = np.round(np.ones(3) * v).astype(np.uint8)
        blue = np.round(np.ones(3) * v).astype(np.uint8)
    else:
        red = np.array([255, 0, 0], dtype=np.uint8)
        blue = np.array([0, 0, 255], dtype=np.uint8)
    return red, blue


def make_params(shape, grayscale=True, thresh=10., count_non_zero=False, background_mask=True,
                max_frame_events=0, flip_x=False, negate_p=False):
    H, W = shape
    red, blue = colour_map(grayscale)
    p = _lib.EcEventsParams()
    p.H, p.W = int(H), int(W)
    p.thresh = float(thresh)
    p.count_non_zero = int(bool(count_non_zero))
    p.background_mask = int(bool(background_mask))
    p.max_frame_events = int(max_frame_events)
    p.flip_x, p.negate_p = int(bool(flip_x)), int(bool(negate_p))
    for c in range(3):
        p.red[c] = int(red[c])
        p.blue[c] = int(blue[c])
    return p


def events_to_frames_device(events, frame_range, shape, grayscale=True, thresh=10.,
                            count_non_zero=False, background_mask=True, return_counts=False,
                            return_stats=False, out=None, max_frame_events=0, flip_x=False,
                            negate_p=False):
    """Batched device entry.

    events:      float32 CUDA tensor [n_total, 4].
    frame_range: int64 CUDA tensor [F, 2] of (begin, end) rows per frame.
    Returns uint8 CUDA tensor [F, H, W, 3] (+ raw, kept int32 [F, H, W, 2] and a
    stats structured array when asked).
    """
    import torch
    dev = _lib.require_gpu()
    H, W = shape
    assert events.is_cuda and events.dtype == torch.float32 and events.is_contiguous()
    assert frame_range.is_cuda and frame_range.dtype == torch.int64 and frame_range.is_contiguous()
    F = int(frame_range.shape[0])
    frames = out if out is not None else torch.empty((F, H, W, 3), dtype=torch.uint8, device=dev)
    raw = kept = stats = None
    if return_counts:
        raw = torch.empty((F, H, W, 2), dtype=torch.int32, device=dev)
        kept = torch.empty((F, H, W, 2), dtype=torch.int32, device=dev)
    if return_stats:
        stats = torch.zeros((F, ctypes.sizeof(_lib.EcFrameStats)), dtype=torch.uint8, device=dev)
    prm = make_params(shape, grayscale, thresh, count_non_zero, background_mask, max_frame_events,
                      flip_x, negate_p)
    rc = _lib.lib().ec_events_to_frames(_lib.ptr(events), _lib.ptr(frame_range), F,
                                        ctypes.byref(prm), _lib.ptr(frames), _lib.ptr(raw),
                                        _lib.ptr(kept), _lib.ptr(stats), _lib.stream_ptr())
    _lib.check(rc, 'ec_events_to_frames')
    res = [frames]
    if return_counts:
        res += [raw, kept]
    if return_stats:
        dt = np.dtype([('sum', '<u8'), ('sumsq', '<u8'), ('nnz', '<u4'), ('max_kept', '<u4'),
                       ('dropped', '<u4'), ('ambiguous', '<u4'), ('thr', '<f8')])
        res.append(stats.cpu().numpy().view(dt).reshape(F))
    return res[0] if len(res) == 1 else tuple(res)


def events2frames(events, split_method, convert_method, shape=(180, 240), **kwargs):
    """Drop-in for the reference's events2frames (vis.py:75-117): numpy in, numpy out."""
    import torch
    grayscale = kwargs.pop('grayscale', True)
    ev = parse_events(events)
    assert split_method == 'event_count'                 # vis.py:90
    if convert_method != 'event_histogram':              # vis.py:113
        raise NotImplementedError(f'{convert_method} not implemented!')
    N = int(kwargs['N'])
    if ev.shape[0] == 0:
        raise IndexError('events2frames: empty event array')
    idx0, idx1 = chunk_bounds(ev.shape[0], N)
    dev = _lib.require_gpu()
    ev_d = torch.from_numpy(ev).to(dev)
    rng = torch.tensor(np.stack([idx0, idx1], 1), dtype=torch.int64, device=dev)
    frames, stats = events_to_frames_device(
        ev_d, rng, shape, grayscale=grayscale, thresh=float(kwargs.get('thresh', 10.)),
        count_non_zero=kwargs.get('count_non_zero', False),
        background_mask=kwargs.get('background_mask', True), return_stats=True,
        max_frame_events=max(b - a for a, b in zip(idx0, idx1)))
    if int(stats['dropped'].sum()) > 0:
        # the reference's bincount/reshape raises on such input (vis.py:11)
        raise ValueError('events2frames: events outside the sensor '
                         f'({int(stats["dropped"].sum())} dropped)')
    return frames.cpu().numpy()


def center_events_device(events, sample_range, resolution):
    """In-place center_events (datasets/utils.py:38-57) for a batch: events float32 CUDA
    [n_total, 4], sample_range int64 CUDA [B, 2]."""
    import torch
    _lib.require_gpu()
    assert events.is_cuda and events.dtype == torch.float32 and events.is_contiguous()
    assert sample_range.is_cuda and sample_range.dtype == torch.int64
    H, W = resolution
    rc = _lib.lib().ec_center_events(_lib.ptr(events), _lib.ptr(sample_range.contiguous()),
                                     int(sample_range.shape[0]), int(H), int(W), _lib.stream_ptr())
    _lib.check(rc, 'ec_center_events')
    return events


def center_events(events, resolution=(180, 240)):
    """Drop-in for datasets/utils.py center_events: numpy [n, 4] in, centred copy out."""
    import torch
    dev = _lib.require_gpu()
    ev = torch.from_numpy(parse_events(events)).to(dev)
    rng = torch.tensor([[0, ev.shape[0]]], dtype=torch.int64, device=dev)
    return center_events_device(ev, rng, resolution).cpu().numpy()
