"""Event-space training augmentation (datasets/caltech.py:153-163 over datasets/utils.py:4-35).

``draw_event_augment`` makes the reference's random draws (same numpy calls, same order, so a seeded
``np.random`` reproduces the reference's stream); ``augment_events_device`` applies them on the GPU
(``ec_augment_events``: optional time flip, shift, drop what left the sensor, optional x flip, survivors
compacted in order).  Dropping events changes the counts, so the result comes with new per-sample
counts and the samples' (unchanged) start offsets for ``Event2ImagePipeline(..., starts=...)``.
"""
import numpy as np
import torch

from . import _lib


def draw_event_augment(n_samples, max_shift=20, flip_time=False, rng=np.random):
    """int32 [n_samples, 4] = (x_shift, y_shift, flip_x, flip_t) per sample."""
    out = np.zeros((n_samples, 4), dtype=np.int32)
    for i in range(n_samples):
        if flip_time:                                                    # caltech.py:155-156, utils.py:28
            out[i, 3] = rng.random() < 0.5
        out[i, 0], out[i, 1] = rng.randint(-max_shift, max_shift + 1, size=(2,))   # utils.py:7
        out[i, 2] = rng.random() < 0.5                                   # utils.py:21
    return out


def augment_events_device(events, n_events, params, resolution):
    """events float32 CUDA [sum n, 4] (samples back to back), n_events per-sample counts, params from
    ``draw_event_augment``.  Returns (events_out [sum n, 4], new_counts list, starts list): sample b's
    surviving events are events_out[starts[b] : starts[b] + new_counts[b]]."""
    dev = _lib.require_gpu()
    assert events.is_cuda and events.dtype == torch.float32 and events.is_contiguous()
    B = len(n_events)
    offs = np.concatenate([[0], np.cumsum(n_events)]).astype(np.int64)
    sr = torch.from_numpy(np.stack([offs[:-1], offs[1:]], 1)).to(dev)
    prm = torch.from_numpy(np.ascontiguousarray(params, dtype=np.int32)).to(dev)
    out = torch.empty_like(events)
    counts = torch.empty((B,), dtype=torch.int64, device=dev)
    H, W = resolution
    rc = _lib.lib().ec_augment_events(_lib.ptr(events), _lib.ptr(sr), B, _lib.ptr(prm), int(H), int(W),
                                      _lib.ptr(out), _lib.ptr(counts), _lib.stream_ptr())
    _lib.check(rc, 'ec_augment_events')
    return out, counts.cpu().tolist(), offs[:-1].tolist()


def augment_events(events, resolution=(180, 240), max_shift=20, flip_time=False, rng=np.random):
    """Drop-in for NCaltech101._augment_events on one sample: numpy [n, 4] in, augmented copy out."""
    dev = _lib.require_gpu()
    ev = torch.from_numpy(np.ascontiguousarray(events, dtype=np.float32)).to(dev)
    prm = draw_event_augment(1, max_shift, flip_time, rng)
    out, counts, _ = augment_events_device(ev, [ev.shape[0]], prm, resolution)
    return out[:counts[0]].cpu().numpy()
