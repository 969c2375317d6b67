"""PyTorch-ROCm custom ops over the C ABI: the ``eventclip_hip::`` namespace.

north_star: "called from Python via PyTorch-ROCm custom ops so models/clip_cls.py's forward/classify
API is a drop-in".  Each op below is a ``torch.library.custom_op`` registered for CUDA (= HIP)
tensors only -- there is no CPU kernel, a CPU tensor raises -- whose body hands device pointers and
torch's current stream to one entry point of libeventclip_hip.so (include/eventclip_hip.h).  The
mirrors of the reference's interface (vis / preprocess / clip / adapter / clip_cls) call these:

    torch.ops.eventclip_hip.events_to_frames   ec_events_to_frames[_packed]    datasets/vis.py:75-117
    torch.ops.eventclip_hip.preprocess         ec_preprocess                   datasets/event2img.py:119-122
    torch.ops.eventclip_hip.vit_encode         ec_vit_encode                   models/clip_cls.py:101
    torch.ops.eventclip_hip.text_encode        ec_text_encode                  models/clip_cls.py:84
    torch.ops.eventclip_hip.adapter_fwd        ec_adapter_forward              models/adapter.py:82-105
    torch.ops.eventclip_hip.classify           ec_classify_v2                   models/clip_cls.py:139-154, :319-343

Weights live in packed C structs owned by the Python modules; an op receives them as an integer
handle into a registry of live modules (tensors-only signatures keep the ops traceable, and the
registered fake kernels give their output shapes without touching the device).
"""
import ctypes
import weakref
from typing import List, Tuple

import numpy as np
import torch
from torch.library import custom_op

from . import _lib

NAMESPACE = 'eventclip_hip'

# ---- handles: id -> weak reference to the module that owns the packed weights ----
_handles = weakref.WeakValueDictionary()


def handle_of(module):
    """Integer handle of a module with packed weights (CLIP, TransformerAdapter)."""
    h = id(module)
    _handles[h] = module
    return h


def _resolve(handle, what):
    m = _handles.get(int(handle))
    if m is None:
        raise RuntimeError(f'{NAMESPACE}::{what}: handle {handle} does not name a live module')
    return m


def _plans():
    from . import preprocess
    return preprocess


# ------------------------------------------------------------------------------------------
# events -> frames
# ------------------------------------------------------------------------------------------
@custom_op(f'{NAMESPACE}::events_to_frames', mutates_args=(), device_types='cuda')
def events_to_frames(events: torch.Tensor, frame_range: torch.Tensor, H: int, W: int, thresh: float,
                     red: List[int], blue: List[int], count_non_zero: bool, background_mask: bool,
                     max_frame_events: int, flip_x: bool, negate_p: bool, float32_stage: bool,
                     total_events: int) -> torch.Tensor:
    """events float32 [n, 4] or packed int64 [n]; frame_range int64 [F, 2] -> uint8 [F, H, W, 3]."""
    from . import vis
    F = int(frame_range.shape[0])
    frames = torch.empty((F, H, W, 3), dtype=torch.uint8, device=events.device)
    prm = _lib.EcEventsParams()
    prm.H, prm.W, prm.thresh = H, W, thresh
    prm.count_non_zero, prm.background_mask = int(count_non_zero), int(background_mask)
    prm.max_frame_events = max_frame_events
    prm.flip_x, prm.negate_p = int(flip_x), int(negate_p)
    prm.float32_stage, prm.total_events = int(float32_stage), total_events
    for c in range(3):
        prm.red[c], prm.blue[c] = red[c], blue[c]
    ws = vis.attach_sort_workspace(prm, events.device, True)   # noqa: F841 (alive over the call)
    packed = vis.is_packed(events)
    entry = _lib.lib().ec_events_to_frames_packed if packed else _lib.lib().ec_events_to_frames
    rc = entry(_lib.ptr(events), _lib.ptr(frame_range), F, ctypes.byref(prm), _lib.ptr(frames), None,
               None, None, _lib.stream_ptr())
    _lib.check(rc, 'ec_events_to_frames')
    return frames


@events_to_frames.register_fake
def _(events, frame_range, H, W, thresh, red, blue, count_non_zero, background_mask, max_frame_events,
      flip_x, negate_p, float32_stage, total_events):
    return events.new_empty((frame_range.shape[0], H, W, 3), dtype=torch.uint8)


# ------------------------------------------------------------------------------------------
# CLIP preprocess
# ------------------------------------------------------------------------------------------
@custom_op(f'{NAMESPACE}::preprocess', mutates_args=(), device_types='cuda')
def preprocess(frames: torch.Tensor, n_px: int, mode: int, patch: int, kpad: int,
               dtype_code: int) -> torch.Tensor:
    """frames uint8 [F, H, W, 3] -> fp32 [F, 3, R, R] (mode EC_PRE_CHW_F32), 16-bit patch rows
    [F, G, kpad] (EC_PRE_PATCHES16) or uint8 [F, R, R, 3] (EC_PRE_HWC_U8)."""
    F, H, W, _ = frames.shape
    host, plan = _plans()._plan(H, W, n_px, frames.device)
    out = frames.new_empty(_pre_shape(F, n_px, mode, patch, kpad), dtype=_pre_dtype(mode, dtype_code))
    rc = _lib.lib().ec_preprocess(_lib.ptr(frames), F, host.ctypes.data, _lib.ptr(plan), _lib.ptr(out),
                                  mode, max(patch, 1), kpad, dtype_code, _lib.stream_ptr())
    _lib.check(rc, 'ec_preprocess')
    return out


def _pre_shape(F, n_px, mode, patch, kpad):
    if mode == _lib.EC_PRE_CHW_F32:
        return (F, 3, n_px, n_px)
    if mode == _lib.EC_PRE_HWC_U8:
        return (F, n_px, n_px, 3)
    return (F, (n_px // patch) ** 2, kpad)


def _pre_dtype(mode, dtype_code):
    if mode == _lib.EC_PRE_CHW_F32:
        return torch.float32
    if mode == _lib.EC_PRE_HWC_U8:
        return torch.uint8
    return torch.float16 if dtype_code == _lib.EC_F16 else torch.bfloat16


@preprocess.register_fake
def _(frames, n_px, mode, patch, kpad, dtype_code):
    return frames.new_empty(_pre_shape(frames.shape[0], n_px, mode, patch, kpad),
                            dtype=_pre_dtype(mode, dtype_code))


# ------------------------------------------------------------------------------------------
# towers
# ------------------------------------------------------------------------------------------
@custom_op(f'{NAMESPACE}::vit_encode', mutates_args=(), device_types='cuda')
def vit_encode(patches: torch.Tensor, clip_handle: int) -> torch.Tensor:
    """patches 16-bit [N, G, kpad] (EC_PRE_PATCHES16 layout) -> fp32 features [N, D]."""
    m = _resolve(clip_handle, 'vit_encode')
    pk = m._pack()
    n = int(patches.shape[0])
    feats = torch.empty((n, m.cfg['embed_dim']), dtype=torch.float32, device=patches.device)
    if n == 0:
        return feats
    chunk = max(1, min(m.chunk, n))
    need = _lib.lib().ec_vit_workspace_bytes(ctypes.byref(pk['vit']), chunk)
    if need > m.workspace_budget:     # keep the scratch bounded (e.g. 336-px inputs)
        chunk = max(1, int(chunk * m.workspace_budget / need))
        need = _lib.lib().ec_vit_workspace_bytes(ctypes.byref(pk['vit']), chunk)
    ws = m._workspace(need, pk['dev'])
    rc = _lib.lib().ec_vit_encode(ctypes.byref(pk['vit']), _lib.ptr(patches), n, _lib.ptr(feats),
                                  _lib.ptr(ws), ws.numel(), chunk, _lib.stream_ptr())
    _lib.check(rc, 'ec_vit_encode')
    return feats


@vit_encode.register_fake
def _(patches, clip_handle):
    return patches.new_empty((patches.shape[0], _resolve(clip_handle, 'vit_encode').cfg['embed_dim']),
                             dtype=torch.float32)


@custom_op(f'{NAMESPACE}::text_encode', mutates_args=(), device_types='cuda')
def text_encode(tokens: torch.Tensor, clip_handle: int) -> torch.Tensor:
    """tokens int32 [K, ctx] -> fp32 [K, D] (not normalised)."""
    m = _resolve(clip_handle, 'text_encode')
    pk = m._pack()
    n = int(tokens.shape[0])
    feats = torch.empty((n, m.cfg['embed_dim']), dtype=torch.float32, device=tokens.device)
    if n == 0:
        return feats
    chunk = max(1, min(512, n))
    need = _lib.lib().ec_text_workspace_bytes(ctypes.byref(pk['text']), chunk)
    ws = m._workspace(need, pk['dev'])
    rc = _lib.lib().ec_text_encode(ctypes.byref(pk['text']), _lib.ptr(tokens), n, _lib.ptr(feats),
                                   _lib.ptr(ws), ws.numel(), chunk, _lib.stream_ptr())
    _lib.check(rc, 'ec_text_encode')
    return feats


@text_encode.register_fake
def _(tokens, clip_handle):
    return tokens.new_empty((tokens.shape[0], _resolve(clip_handle, 'text_encode').cfg['embed_dim']),
                            dtype=torch.float32)


# ------------------------------------------------------------------------------------------
# adapter, classifier tail
# ------------------------------------------------------------------------------------------
@custom_op(f'{NAMESPACE}::adapter_fwd', mutates_args=(), device_types='cuda')
def adapter_fwd(feats: torch.Tensor, row_idx: torch.Tensor, adapter_handle: int) -> torch.Tensor:
    """feats fp32 [Nv, C] (compact valid views), row_idx int32 [B, T] (-1 = padded) -> [B, T, C]."""
    a = _resolve(adapter_handle, 'adapter_fwd')
    pk = a._pack()
    B, T = row_idx.shape
    out = torch.empty((B, T, a.in_dim), dtype=torch.float32, device=feats.device)
    rc = _lib.lib().ec_adapter_forward(ctypes.byref(pk['w']), _lib.ptr(feats), _lib.ptr(row_idx), B, T,
                                       _lib.ptr(out), _lib.stream_ptr())
    _lib.check(rc, 'ec_adapter_forward')
    return out


@adapter_fwd.register_fake
def _(feats, row_idx, adapter_handle):
    return feats.new_empty((row_idx.shape[0], row_idx.shape[1], feats.shape[-1]))


# prepared text planes of ec_classify_prep_text, keyed on the identity and version of the text_t tensor: the text
# features are constant across batches (cached prompts) or change once per optimiser step (learned parameter, a new
# tensor each time: clip_cls.FSCLIPClassifier._text_transposed), so the transposed hi + lo planes are built once
_TEXT_PLANES = {}
_TEXT_PLANES_MAX = 8


def _text_planes(text_t):
    C, K = text_t.shape
    key = (text_t.data_ptr(), text_t._version, C, K, text_t.device.index)
    hit = _TEXT_PLANES.get(key)
    if hit is not None and hit[1]() is text_t:
        return hit[0]
    import weakref
    ws = torch.empty(max(int(_lib.lib().ec_classify_text_bytes(C, K)), 256), dtype=torch.uint8, device=text_t.device)
    rc = _lib.lib().ec_classify_prep_text(_lib.ptr(text_t), C, K, _lib.ptr(ws), ws.numel(), _lib.stream_ptr())
    _lib.check(rc, 'ec_classify_prep_text')
    for k in [k for k, v in _TEXT_PLANES.items() if v[1]() is None]:        # tensors that are gone
        del _TEXT_PLANES[k]
    while len(_TEXT_PLANES) >= _TEXT_PLANES_MAX:
        del _TEXT_PLANES[next(iter(_TEXT_PLANES))]
    _TEXT_PLANES[key] = (ws, weakref.ref(text_t))
    return ws


@custom_op(f'{NAMESPACE}::classify', mutates_args=(), device_types='cuda')
def classify(feats: torch.Tensor, row_idx: torch.Tensor, text_t: torch.Tensor, logit_scale: float,
             agg: int, normalize: bool) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """feats fp32 [Nv, C], row_idx int32 [B, T], text_t fp32 [C, K] -> (full_logits [B, T, K],
    logits [B, K], probs [B, K])."""
    B, T = row_idx.shape
    C, K = text_t.shape
    full = torch.empty((B, T, K), dtype=torch.float32, device=feats.device)
    logits = torch.empty((B, K), dtype=torch.float32, device=feats.device)
    probs = torch.empty((B, K), dtype=torch.float32, device=feats.device)
    n_rows = int(feats.shape[0])
    text_ws = _text_planes(text_t)
    ws = torch.empty(max(int(_lib.lib().ec_classify_v2_workspace_bytes(n_rows, C, K)), 256), dtype=torch.uint8, device=feats.device)
    rc = _lib.lib().ec_classify_v2(_lib.ptr(feats), n_rows, _lib.ptr(row_idx), _lib.ptr(text_ws), B, T, C, K,
                                   logit_scale, agg, int(normalize), _lib.ptr(full), _lib.ptr(logits),
                                   _lib.ptr(probs), _lib.ptr(ws), ws.numel(), _lib.stream_ptr())
    _lib.check(rc, 'ec_classify_v2')
    return full, logits, probs


@classify.register_fake
def _(feats, row_idx, text_t, logit_scale, agg, normalize):
    B, T = row_idx.shape
    K = text_t.shape[1]
    return feats.new_empty((B, T, K)), feats.new_empty((B, K)), feats.new_empty((B, K))


OPS = ('events_to_frames', 'preprocess', 'vit_encode', 'text_encode', 'adapter_fwd', 'classify')
