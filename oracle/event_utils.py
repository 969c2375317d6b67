"""Oracle for the event-space transforms that run at test time (TEST INFRASTRUCTURE).

Restates /root/reference/datasets/utils.py: center_events (:38-57, called for every
N-Caltech / N-ImageNet sample at caltech.py:176), random_flip_events_along_x with p=1
(:18-23) and random_time_flip_events with p=1 (:26-35) -- the two flips the reference
composes into its 4-view test-time augmentation (datasets/event2img.py:94-112).
Pinned by tests/golden/event_utils.npz, produced by importing the reference's utils.py
(tools/make_golden_event_utils.py).
"""
import numpy as np


def center_events(events, resolution=(180, 240)):
    """utils.py:38-57 (in place on a float array, like the reference)."""
    events[:, 2] -= events[:, 2].min()
    H, W = resolution
    x_min, x_max = events[:, 0].min(), events[:, 0].max()
    y_min, y_max = events[:, 1].min(), events[:, 1].max()
    x_shift = ((x_max + x_min + 1.) - W) // 2.
    y_shift = ((y_max + y_min + 1.) - H) // 2.
    events[:, 0] -= x_shift
    events[:, 1] -= y_shift
    return events


def hflip_events(events, resolution=(180, 240)):
    """utils.py:18-23 with p = 1."""
    H, W = resolution
    events[:, 0] = W - 1 - events[:, 0]
    return events


def tflip_events(events):
    """utils.py:26-35 with p = 1: reverse order, t -> t[0] - t, p -> -p."""
    events = np.ascontiguousarray(np.flip(events, axis=0))
    events[:, 2] = events[0, 2] - events[:, 2]
    events[:, 3] = -events[:, 3]
    return events


def tta_views(events, resolution):
    """event2img.py:97-103: [events, h-flip, t-flip, h+t-flip]."""
    h = hflip_events(events.copy(), resolution)
    return [events, h, tflip_events(events.copy()), tflip_events(h.copy())]


def load_event_npz(path_or_records):
    """N-ImageNet reader, datasets/imagenet.py:8-27: structured (x, y, t [us], p) records ->
    float64 [n, 4] with t in seconds and polarity 0 mapped to -1 when no negative polarity is
    present.  Pinned by tests/golden/ingest.npz (tools/make_golden_ingest.py runs the reference's
    load_event on the same records)."""
    rec = np.load(path_or_records)['event_data'] if isinstance(path_or_records, str) else path_or_records
    event = np.stack([rec['x'], rec['y'], rec['t'], rec['p'].astype(np.uint8)], 1).astype(float)
    event[:, 2] /= 1e6
    if event[:, 3].min() >= -0.5:
        event[:, 3][event[:, 3] <= 0.5] = -1
    return event


def packed_fields(events):
    """What the 8-byte packed form (include/eventclip_hip.h) must carry for float events
    [n, 4]: parse_events' truncated x, y (vis.py:50), the polarity code 0 / 1 / 2 for
    p == 0 / p > 0 / p < 0 (vis.py:10,12) and t in whole microseconds."""
    ev = np.asarray(events, dtype=np.float32)
    p = ev[:, 3].astype(np.int32)
    code = np.where(p == 0, 0, np.where(p > 0, 1, 2))
    t_us = np.clip(np.rint(ev[:, 2].astype(np.float64) * 1e6), 0, (1 << 30) - 1).astype(np.int64)
    return ev[:, 0].astype(np.int32), ev[:, 1].astype(np.int32), code, t_us


def augment_events(events, params, resolution):
    """NCaltech101._augment_events (caltech.py:153-163) with the draws given: params =
    (x_shift, y_shift, flip_x, flip_t).  utils.py:26-35, :4-15, :18-23 in that order."""
    dx, dy, flip_x, flip_t = (int(v) for v in params)
    H, W = resolution
    ev = np.array(events, copy=True)
    if flip_t:
        ev = np.ascontiguousarray(np.flip(ev, axis=0))
        ev[:, 2] = ev[0, 2] - ev[:, 2]
        ev[:, 3] = -ev[:, 3]
    ev[:, 0] += dx
    ev[:, 1] += dy
    keep = (ev[:, 0] >= 0) & (ev[:, 0] < W) & (ev[:, 1] >= 0) & (ev[:, 1] < H)
    ev = ev[keep]
    if flip_x:
        ev[:, 0] = W - 1 - ev[:, 0]
    return ev
