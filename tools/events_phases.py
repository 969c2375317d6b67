"""Phase times of events_pack10_kernel per frame (diagnostic build: python -m eventclip_amd.build --diag).
Stamps: 0 start, 1 bins zeroed, 2 events read + binned, 3 pass 1 + threshold, 4 pass 2, 5 LUT, 6 pass 3.
Run on the GPU box: EVENTCLIP_HIP_LIB=eventclip_amd/libeventclip_hip_diag.so python tools/events_phases.py"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('EVENTCLIP_HIP_LIB', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                       'eventclip_amd', 'libeventclip_hip_diag.so'))
from eventclip_amd import _lib, vis  # noqa: E402
from eventclip_amd.synthetic import make_events  # noqa: E402

shape, n, frames = (180, 240), 20000, 2560
packed = len(sys.argv) > 1 and sys.argv[1] == 'packed'
ev = np.concatenate([make_events(n, shape, seed=i) for i in range(8)] * (frames // 8))
rng = torch.tensor([[i * n, (i + 1) * n] for i in range(frames)], dtype=torch.int64).cuda()
e = torch.from_numpy(vis.pack_events(ev).view(np.int64) if packed else ev).cuda()
out = torch.empty((frames, *shape, 3), dtype=torch.uint8, device='cuda')
for _ in range(3):
    vis.events_to_frames_device(e, rng, shape, grayscale=False, out=out, max_frame_events=n)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["EVENTCLIP_HIP_LIB"])
buf = np.zeros((frames, 8), dtype=np.uint64)
lib.ec_events_phase_times.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.ec_events_phase_times.restype = ctypes.c_int
assert lib.ec_events_phase_times(buf.ctypes.data, frames) == 0
t = buf[:, :7].astype(np.float64) / 100.0          # wall_clock64: 100 MHz -> us
t0 = t[:, 0].min()
d = np.diff(t, axis=1)
names = ['zero', 'read+bin', 'pass1+thr', 'pass2', 'lut', 'pass3']
print('launch span (first start -> last end): %.1f us' % (t[:, 6].max() - t0))
for k, nm in enumerate(names):
    print(f'{nm:10s} mean {d[:, k].mean():7.2f} us   p10 {np.percentile(d[:, k], 10):7.2f}   p90 {np.percentile(d[:, k], 90):7.2f}')
print('frame total mean %.2f us' % (t[:, 6] - t[:, 0]).mean())
# per round of 256 frames in start order
order = np.argsort(t[:, 0])
for r in range(0, frames, 256):
    idx = order[r:r + 256]
    print('round %2d: start %.1f..%.1f  read+bin %.1f  rest %.1f' % (
        r // 256, t[idx, 0].min() - t0, t[idx, 0].max() - t0, d[idx, 1].mean(), (t[idx, 6] - t[idx, 2]).mean()))
