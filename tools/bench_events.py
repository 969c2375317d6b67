"""events -> frames kernel alone at the three dataset geometries (run on the GPU box)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eventclip_amd import vis  # noqa: E402
from eventclip_amd.synthetic import make_events  # noqa: E402

for name, shape, n, frames in (('n_caltech', (180, 240), 20000, 2560), ('n_cars', (100, 120), 12500, 512),
                               ('n_imagenet', (480, 640), 70000, 512), ('n_imagenet', (480, 640), 70000, 2560)):
    uniq = 8
    ev = np.concatenate([make_events(n, shape, seed=i) for i in range(uniq)] * (frames // uniq))
    rng = torch.tensor([[i * n, (i + 1) * n] for i in range(frames)], dtype=torch.int64).cuda()
    for packed in (False, True):
        e = torch.from_numpy(vis.pack_events(ev).view(np.int64) if packed else ev).cuda()
        out = torch.empty((frames, *shape, 3), dtype=torch.uint8, device='cuda')
        for _ in range(3):
            vis.events_to_frames_device(e, rng, shape, grayscale=False, out=out, max_frame_events=n)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            vis.events_to_frames_device(e, rng, shape, grayscale=False, out=out, max_frame_events=n)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        alg = frames * ((8 if packed else 16) * n + 3 * shape[0] * shape[1])
        print(f'{name:10s} {"packed" if packed else "float "}: {ms:7.3f} ms / {frames} frames = '
              f'{frames / ms * 1e3:9.0f} frames/s, {alg / ms / 1e6:7.1f} GB/s algorithmic', flush=True)
