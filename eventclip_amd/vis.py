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


# Which numpy the float stage of make_event_histogram (vis.py:27-39) agrees with when the caller does not say.
# The reference pins numpy 1.25.2 (/root/reference/environment.yml:49), whose value-based casting keeps
# `hist.astype(np.float32) / hist.max()` -- and everything after it -- in float32; under numpy >= 2 (NEP 50, this
# image) the same line promotes to float64.  The two differ by 1 LSB at exact .5 ties only.  Default: the pinned
# reference environment; quantize_args['float_stage'] = 'float64' gives what this image's numpy makes of the
# reference (both are pinned by every events fixture).
DEFAULT_FLOAT_STAGE = 'float32'


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
                max_frame_events=0, flip_x=False, negate_p=False, float_stage=DEFAULT_FLOAT_STAGE,
                total_events=0):
    H, W = shape
    red, blue = colour_map(grayscale)
    p = _lib.EcEventsParams()
    p.H, p.W = int(H), int(W)
    p.thresh = float(thresh)
    p.count_non_zero = int(bool(count_non_zero))
    p.background_mask = int(bool(background_mask))
    p.max_frame_events = int(max_frame_events)
    p.flip_x, p.negate_p = int(bool(flip_x)), int(bool(negate_p))
    if float_stage not in ('float64', 'float32'):
        raise ValueError(f'float_stage {float_stage!r}: float64 (numpy >= 2) or float32 (numpy 1.x)')
    p.float32_stage = int(float_stage == 'float32')
    p.total_events = int(total_events)
    for c in range(3):
        p.red[c] = int(red[c])
        p.blue[c] = int(blue[c])
    return p


_SORT_WS = {}   # device index -> uint8 CUDA tensor (grown as needed, reused by every call)


def attach_sort_workspace(prm, device, enable=True):
    """Long frames (N-ImageNet: 70 000 events on 480 x 640) are bucketed by row band in a device
    scratch instead of being re-read for every band pass; the library says how much pays off."""
    import torch
    need = int(_lib.lib().ec_events_sort_workspace_bytes(ctypes.byref(prm))) if enable else 0
    if need <= 0:
        return None
    key = device.index if device.index is not None else torch.cuda.current_device()
    ws = _SORT_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = _SORT_WS[key] = torch.empty((need,), dtype=torch.uint8, device=device)
    prm.sort_workspace, prm.sort_workspace_bytes = ws.data_ptr(), ws.numel()
    return ws


# ---- packed 8-byte events (include/eventclip_hip.h: x | y << 16 | code << 32 | t_us << 34) ----
PACKED_T_MAX = (1 << 30) - 1


def is_packed(events):
    """Packed events travel as 1-D int64 (torch) / uint64 or int64 (numpy) arrays."""
    return events.ndim == 1 and str(events.dtype).split('.')[-1] in ('int64', 'uint64')


def pack_events(events):
    """Host packer: float [n, 4] (x, y, t [s], p) -> uint64 [n].  Same bits as ec_pack_events.
    Raises when a coordinate is not an integer in [0, 65535] (the float path flips x before it
    truncates, datasets/utils.py:22, so such events have no packed equivalent)."""
    ev = np.asarray(events)
    if ev.dtype.names:   # structured (x, y, t, p) record array
        ev = np.stack([ev['x'], ev['y'], ev['t'], ev['p']], 1)
    ev = ev.astype(np.float32, copy=False)
    x, y = ev[:, 0].astype(np.int32), ev[:, 1].astype(np.int32)     # vis.py:50
    if not ((x == ev[:, 0]) & (y == ev[:, 1]) & (x >= 0) & (y >= 0) & (x < 65536) & (y < 65536)).all():
        raise ValueError('pack_events: coordinates must be integers in [0, 65535]')
    p = ev[:, 3].astype(np.int32)
    code = np.where(p == 0, 0, np.where(p > 0, 1, 2)).astype(np.uint64)
    t_us = np.clip(np.rint(ev[:, 2].astype(np.float64) * 1e6), 0, PACKED_T_MAX).astype(np.uint64)
    return (x.astype(np.uint64) | (y.astype(np.uint64) << np.uint64(16)) | (code << np.uint64(32)) |
            (t_us << np.uint64(34)))


def pack_structured(x, y, t_us, p):
    """Packed events straight from integer fields (N-ImageNet's event_data, imagenet.py:8-27:
    t already in microseconds, polarity 0/1 where 0 means negative unless negatives are present)."""
    x, y = np.asarray(x).astype(np.int64), np.asarray(y).astype(np.int64)
    if x.size and (x.min() < 0 or y.min() < 0 or x.max() > 65535 or y.max() > 65535):
        raise ValueError('pack_structured: coordinates must lie in [0, 65535]')
    p = np.asarray(p).astype(np.uint8).astype(np.float64)   # imagenet.py:15 (uint8 cast first)
    if p.size and p.min() >= -0.5:                         # imagenet.py:24-25
        p = np.where(p <= 0.5, -1., p)
    code = np.where(p == 0, 0, np.where(p > 0, 1, 2)).astype(np.uint64)
    t = np.clip(np.asarray(t_us).astype(np.int64), 0, PACKED_T_MAX).astype(np.uint64)
    return (x.astype(np.uint64) | (y.astype(np.uint64) << np.uint64(16)) | (code << np.uint64(32)) |
            (t << np.uint64(34)))


def unpack_events(packed):
    """uint64 [n] -> float32 [n, 4] (x, y, t [s], p in {-1, 0, +1})."""
    e = np.asarray(packed).astype(np.uint64)
    code = (e >> np.uint64(32)) & np.uint64(3)
    p = np.where(code == 0, 0., np.where(code == 1, 1., -1.))
    return np.stack([(e & np.uint64(0xffff)).astype(np.float64),
                     ((e >> np.uint64(16)) & np.uint64(0xffff)).astype(np.float64),
                     (e >> np.uint64(34)).astype(np.float64) / 1e6, p], 1).astype(np.float32)


def pack_events_device(events, return_bad=False):
    """float32 CUDA [n, 4] -> packed int64 CUDA [n] (ec_pack_events)."""
    import torch
    dev = _lib.require_gpu()
    assert events.is_cuda and events.dtype == torch.float32 and events.is_contiguous()
    n = int(events.shape[0])
    out = torch.empty((n,), dtype=torch.int64, device=dev)
    bad = torch.zeros((1,), dtype=torch.int32, device=dev)
    rc = _lib.lib().ec_pack_events(_lib.ptr(events), n, _lib.ptr(out), _lib.ptr(bad), _lib.stream_ptr())
    _lib.check(rc, 'ec_pack_events')
    return (out, int(bad.item())) if return_bad else out


def events_to_frames_device(events, frame_range, shape, grayscale=True, thresh=10.,
                            count_non_zero=False, background_mask=True, return_counts=False,
                            return_stats=False, out=None, max_frame_events=0, flip_x=False,
                            negate_p=False, sort_workspace=True, float_stage=DEFAULT_FLOAT_STAGE,
                            total_events=0):
    """Batched device entry.

    events:      float32 CUDA tensor [n_total, 4], or packed events: int64 CUDA tensor [n_total]
                 (pack_events / pack_events_device, layout in include/eventclip_hip.h).
    frame_range: int64 CUDA tensor [F, 2] of (begin, end) rows per frame.
    float_stage: 'float32' runs vis.py:27-39 as the reference's pinned numpy 1.25 does (the default,
                 DEFAULT_FLOAT_STAGE), 'float64' as numpy >= 2 does.
    total_events: sum of the frame lengths when known (profiling: the launch's algorithmic bytes).
    Returns uint8 CUDA tensor [F, H, W, 3] (+ raw, kept int32 [F, H, W, 2] and a
    stats structured array when asked).
    """
    import torch
    dev = _lib.require_gpu()
    H, W = shape
    packed = is_packed(events)
    assert events.is_cuda and events.is_contiguous()
    assert packed or (events.dtype == torch.float32 and events.dim() == 2 and events.shape[1] == 4)
    assert frame_range.is_cuda and frame_range.dtype == torch.int64 and frame_range.is_contiguous()
    F = int(frame_range.shape[0])
    if out is None and not return_counts and not return_stats and sort_workspace:
        # the production form: the registered custom op (eventclip_hip::events_to_frames)
        from . import torch_ops  # noqa: F401  (registers the namespace)
        red, blue = colour_map(grayscale)
        if float_stage not in ('float64', 'float32'):
            raise ValueError(f'float_stage {float_stage!r}: float64 (numpy >= 2) or float32 (numpy 1.x)')
        return torch.ops.eventclip_hip.events_to_frames(
            events, frame_range, int(H), int(W), float(thresh), [int(v) for v in red],
            [int(v) for v in blue], bool(count_non_zero), bool(background_mask), int(max_frame_events),
            bool(flip_x), bool(negate_p), float_stage == 'float32', int(total_events))
    # debug form (raw / kept counts, per-frame statistics, caller-owned output): straight to the C ABI
    frames = out if out is not None else torch.empty((F, H, W, 3), dtype=torch.uint8, device=dev)
    raw = kept = stats = None
    if return_counts:
        raw = torch.empty((F, H, W, 2), dtype=torch.int32, device=dev)
        kept = torch.empty((F, H, W, 2), dtype=torch.int32, device=dev)
    if return_stats:
        stats = torch.zeros((F, ctypes.sizeof(_lib.EcFrameStats)), dtype=torch.uint8, device=dev)
    prm = make_params(shape, grayscale, thresh, count_non_zero, background_mask, max_frame_events,
                      flip_x, negate_p, float_stage, total_events)
    ws = attach_sort_workspace(prm, events.device, sort_workspace)   # noqa: F841 (kept alive over the call)
    entry = _lib.lib().ec_events_to_frames_packed if packed else _lib.lib().ec_events_to_frames
    rc = entry(_lib.ptr(events), _lib.ptr(frame_range), F, ctypes.byref(prm), _lib.ptr(frames),
               _lib.ptr(raw), _lib.ptr(kept), _lib.ptr(stats), _lib.stream_ptr())
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
        max_frame_events=max(b - a for a, b in zip(idx0, idx1)),
        float_stage=kwargs.get('float_stage', DEFAULT_FLOAT_STAGE),
        total_events=sum(b - a for a, b in zip(idx0, idx1)))
    if int(stats['dropped'].sum()) > 0:
        # the reference's bincount/reshape raises on such input (vis.py:11)
        raise ValueError('events2frames: events outside the sensor '
                         f'({int(stats["dropped"].sum())} dropped)')
    return frames.cpu().numpy()


def center_events_device(events, sample_range, resolution):
    """In-place center_events (datasets/utils.py:38-57) for a batch: events float32 CUDA
    [n_total, 4] (or packed int64 [n_total]), sample_range int64 CUDA [B, 2]."""
    import torch
    _lib.require_gpu()
    packed = is_packed(events)
    assert events.is_cuda and events.is_contiguous() and (packed or events.dtype == torch.float32)
    assert sample_range.is_cuda and sample_range.dtype == torch.int64
    H, W = resolution
    entry = _lib.lib().ec_center_events_packed if packed else _lib.lib().ec_center_events
    sample_range = sample_range.contiguous()
    rc = entry(_lib.ptr(events), _lib.ptr(sample_range), int(sample_range.shape[0]),
               int(H), int(W), _lib.stream_ptr())
    _lib.check(rc, 'ec_center_events')
    return events


def center_events(events, resolution=(180, 240)):
    """Drop-in for datasets/utils.py center_events: numpy [n, 4] in, centred copy out."""
    import torch
    dev = _lib.require_gpu()
    ev = torch.from_numpy(parse_events(events)).to(dev)
    rng = torch.tensor([[0, ev.shape[0]]], dtype=torch.int64, device=dev)
    return center_events_device(ev, rng, resolution).cpu().numpy()
