"""Events -> model-ready views on the MI355X: device-side mirror of the reference's
``Event2ImageDataset`` (datasets/event2img.py:14-145) plus the DataLoader collate.

The reference converts ONE sample per ``__getitem__`` in CPU worker processes:
``events2frames`` (:118) -> PIL -> ``self.transforms`` per frame (:119-122) ->
``_subsample_imgs`` (:80-92, pad with zero tensors / random subset to ``max_imgs``)
and the loader stacks samples into ``img [B, T, 3, R, R]``, ``valid_mask [B, T]``.
Here a whole batch goes through two kernels (events->frames, preprocess) with the
chunk bookkeeping (``split_event_count``, ``max_imgs``, padding, masks) done on the
host from event counts alone.  No CPU fallback.
"""
import copy

import numpy as np
import torch

from . import _lib
from . import vis
from .preprocess import preprocess_frames


class Event2ImagePipeline:
    """Batched ``_event2img`` (event2img.py:114-128).

    resolution, max_n: the event dataset's constants (caltech.py:52-58 etc.).
    quantize_args: same dict as the reference's configs (``max_imgs``, ``N``,
        ``split_method``, ``convert_method``, ``grayscale``, ``count_non_zero``,
        ``background_mask``).
    n_px: CLIP input resolution.  patch/kpad/dtype: when given, ``__call__`` emits the
        16-bit im2col rows for ``CLIP.encode_patches`` instead of fp32 images.
    """

    def __init__(self, resolution, max_n, quantize_args, n_px=224, patch=None, kpad=None,
                 dtype=torch.float16, generator=None, augment=False):
        qa = copy.deepcopy(quantize_args)
        self.resolution = tuple(resolution)
        self.split_method = qa['split_method']
        self.event_rep = qa['convert_method']
        assert self.split_method == 'event_count'                       # event2img.py:69
        if self.event_rep != 'event_histogram':                          # vis.py:113
            raise NotImplementedError(f'{self.event_rep} not implemented!')
        self.N = int(qa['N'])
        max_imgs = round(max_n / self.N)                                 # event2img.py:70
        max_max_imgs = qa.pop('max_imgs', 10)                            # event2img.py:71
        self.max_imgs = max(min(max_imgs, max_max_imgs), 1)              # event2img.py:72
        self.grayscale = qa.get('grayscale', True)
        self.thresh = float(qa.get('thresh', 10.))
        self.count_non_zero = bool(qa.get('count_non_zero', False))
        self.background_mask = bool(qa.get('background_mask', True))
        # not a key of the reference's configs: which numpy the frames should agree with (vis.py:27)
        self.float_stage = qa.get('float_stage', 'float64')
        self.n_px, self.patch, self.kpad, self.dtype = int(n_px), patch, kpad, dtype
        self.generator = generator
        self.strict = True   # raise on events outside the sensor, as the reference does
        # training-time RandAugment on the frames (event2img.py:34-42, :120-121): two ops per sample,
        # bicubic, white fill on masked-background frames, black otherwise
        self.augment = bool(augment)
        if self.augment:
            from .randaugment import RandAugment
            self.augmentation = RandAugment(
                num_ops=2, interpolation='bicubic',
                fill=[255, 255, 255] if self.background_mask else [0, 0, 0])

    # ---- host bookkeeping: which event rows make which view ----
    def plan(self, n_events, tflip=False, starts=None):
        """n_events: per-sample event counts.  Returns (frame_range int64 [Fv, 2],
        row_idx int32 [B, T], valid_mask bool [B, T]) as CPU tensors; row_idx[b, t] is the
        compact frame number of view t of sample b, or -1 for a padded view.
        tflip: chunk the time-reversed stream (utils.py:26-35): chunk [a, b) of the reversed
        order covers rows [n - b, n - a) of the stored order (a histogram ignores order).
        starts: first event row of every sample when the samples are not back to back (after
        eventclip_amd.augment.augment_events_device has dropped events)."""
        T = self.max_imgs
        B = len(n_events)
        ranges, row_idx = [], np.full((B, T), -1, dtype=np.int32)
        off = 0
        for b, n in enumerate(n_events):
            n = int(n)
            if starts is not None:
                off = int(starts[b])
            if n <= 0:
                raise IndexError('sample with no events (the reference resamples these upstream, '
                                 'caltech.py:181-182)')
            idx0, idx1 = vis.chunk_bounds(n, self.N)
            sel = list(range(len(idx0)))
            if len(sel) > T:                                             # event2img.py:83-86
                sel = torch.randperm(len(sel), generator=self.generator)[:T].tolist()
            for t, f in enumerate(sel):                                  # event2img.py:87-91
                row_idx[b, t] = len(ranges)
                if tflip:
                    ranges.append((off + n - idx1[f], off + n - idx0[f]))
                else:
                    ranges.append((off + idx0[f], off + idx1[f]))
            off += n
        fr = torch.tensor(ranges, dtype=torch.int64).reshape(-1, 2)
        ri = torch.from_numpy(row_idx)
        return fr, ri, ri >= 0

    def frames(self, events, frame_range, hflip=False, tflip=False, total_events=0):
        """uint8 [Fv, H, W, 3] for the planned views (vis.events2frames, batched)."""
        out = vis.events_to_frames_device(
            events, frame_range, self.resolution, grayscale=self.grayscale, thresh=self.thresh,
            count_non_zero=self.count_non_zero, background_mask=self.background_mask,
            return_stats=self.strict, max_frame_events=self.N, flip_x=hflip, negate_p=tflip,
            float_stage=self.float_stage, total_events=total_events)
        if self.strict:
            frames, stats = out
            if int(stats['dropped'].sum()) > 0:
                raise ValueError('events outside the sensor '
                                 f'({int(stats["dropped"].sum())} dropped; vis.py:11 would raise)')
            return frames
        return out

    @staticmethod
    def _concat(samples, dev):
        """Per-sample arrays -> one device tensor + counts.  Samples are float [n_i, 4] (or the
        reference's dict form) or packed uint64 / int64 [n_i] (vis.pack_events)."""
        n_events = [int(e.shape[0]) for e in samples]
        host = [e.cpu().numpy() if torch.is_tensor(e) else e for e in samples]
        if all(not isinstance(e, dict) and vis.is_packed(e) for e in host):
            cat = np.concatenate([np.asarray(e).view(np.int64) for e in host])
        else:
            cat = np.concatenate([vis.parse_events(e) for e in host], axis=0)
        return torch.from_numpy(cat).to(dev), n_events

    def __call__(self, events, n_events=None, hflip=False, tflip=False, center=False, starts=None):
        """events: list of per-sample float32 [n_i, 4] arrays/tensors (or packed uint64 [n_i],
        vis.pack_events), or one CUDA tensor [sum n_i, 4] (packed: int64 [sum n_i]) with
        ``n_events`` giving the per-sample counts.
        hflip / tflip: the test-time-augmentation views of event2img.py:94-112;
        center: apply center_events (utils.py:38-57, in place) first, as the N-Caltech /
        N-ImageNet readers do (caltech.py:176).

        Returns a dict with ``valid_mask`` [B, T] (CUDA bool), ``row_idx`` [B, T] (CUDA
        int32) and either ``patches`` [Fv, G, kpad] (fused path) or ``img``
        [B, T, 3, R, R] float32 (the reference's batch layout, padded views all-zero)."""
        dev = _lib.require_gpu()
        if isinstance(events, (list, tuple)):
            events, n_events = self._concat(events, dev)
        assert events.is_cuda and n_events is not None
        assert vis.is_packed(events) or events.dtype == torch.float32
        if center:
            o0 = np.asarray(starts if starts is not None else np.concatenate([[0], np.cumsum(n_events)[:-1]]))
            offs = np.stack([o0, o0 + np.asarray(n_events)], 1)
            sr = torch.tensor(offs, dtype=torch.int64, device=dev)
            vis.center_events_device(events, sr, self.resolution)
        fr, ri, vm = self.plan(n_events, tflip=tflip, starts=starts)
        fr_d = fr.to(dev)
        frames = self.frames(events, fr_d, hflip=hflip, tflip=tflip,
                             total_events=int((fr[:, 1] - fr[:, 0]).sum()))
        out = dict(valid_mask=vm.to(dev), row_idx=ri.to(dev))
        if self.augment:
            frames = self._augment_frames(frames, ri)
        if self.patch:
            out['patches'] = preprocess_frames(frames, self.n_px, mode='patches', patch=self.patch,
                                               kpad=self.kpad, dtype=self.dtype)
        else:
            chw = preprocess_frames(frames, self.n_px, mode='chw')
            B, T = ri.shape
            img = torch.zeros((B, T, 3, self.n_px, self.n_px), dtype=torch.float32, device=dev)
            img[out['valid_mask']] = chw                                 # event2img.py:88-91
            out['img'] = img
        return out


    def _augment_frames(self, frames, row_idx):
        """One RandAugment draw per sample, in sample order (the reference draws inside
        ``__getitem__``, once per sample), the same ops for all of the sample's views."""
        from .randaugment import apply_ops
        per_frame = [None] * int(frames.shape[0])
        for b in range(row_idx.shape[0]):
            self.augmentation.randomize_ops(self.resolution)
            ops, self.augmentation.cur_ops = self.augmentation.cur_ops, None
            for r in row_idx[b].tolist():
                if r >= 0:
                    per_frame[r] = ops
        return apply_ops(frames, per_frame, self.augmentation._fill())

    def tta(self, events, n_events=None):
        """The four views of _load_tta_data (event2img.py:94-112): identity, h-flip, t-flip,
        h+t-flip, as a list of batches in that order."""
        if isinstance(events, (list, tuple)):
            events, n_events = self._concat(events, _lib.require_gpu())
        return [self(events, n_events, hflip=h, tflip=t)
                for h, t in ((False, False), (True, False), (False, True), (True, True))]


def build_event2img_pipeline(params, resolution, max_n, clip_model=None):
    """Counterpart of build_event2img_dataset (event2img.py:148-156) for the device path."""
    kw = {}
    n_px = 224
    if clip_model is not None:
        c = clip_model.cfg
        n_px = c['image_size']
        kw = dict(patch=c['patch'], kpad=clip_model.kpad, dtype=clip_model.compute_dtype)
    return Event2ImagePipeline(resolution, max_n, params.quantize_args, n_px=n_px, **kw)
