"""CLIP preprocess kernel alone (uint8 frames -> 16-bit patch rows / fp32 CHW) at the dataset geometries.
Run on the GPU box: python tools/bench_preprocess.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eventclip_amd import preprocess  # noqa: E402

for name, shape, n_px, patch, frames in (('n_caltech -> 224 / 14', (180, 240), 224, 14, 2560), ('n_cars -> 224 / 14', (100, 120), 224, 14, 2560),
                                         ('n_imagenet -> 224 / 14', (480, 640), 224, 14, 1024), ('n_imagenet -> 336 / 14', (480, 640), 336, 14, 512)):
    g = torch.Generator(device='cuda').manual_seed(1)
    fr = torch.randint(0, 256, (frames, *shape, 3), dtype=torch.uint8, device='cuda', generator=g)
    klo = ((3 * patch * patch + 63) // 64) * 64
    kpad = ((2 * 3 * patch * patch + 63) // 64) * 64
    G = (n_px // patch) ** 2
    for mode, out, nbytes in (('patches', torch.empty((frames, G, kpad), dtype=torch.float16, device='cuda'), frames * G * kpad * 2),
                              ('chw', torch.empty((frames, 3, n_px, n_px), dtype=torch.float32, device='cuda'), frames * 3 * n_px * n_px * 4)):
        kw = dict(patch=patch, kpad=kpad) if mode == 'patches' else {}
        for _ in range(3):
            preprocess.preprocess_frames(fr, n_px, mode, out=out, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            preprocess.preprocess_frames(fr, n_px, mode, out=out, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        alg = fr.numel() + nbytes
        print(f'{name:24s} {mode:8s}: {ms:7.3f} ms / {frames} frames, {alg / ms / 1e6:7.1f} GB/s algorithmic ({alg / ms / 8e9:.3f} of HBM peak)', flush=True)
    del fr
