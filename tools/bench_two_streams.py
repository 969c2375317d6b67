"""Experiment: the image tower over one 2560-frame chunk on ONE stream against the same frames as
two independent halves on TWO streams with separate workspaces, so that one half's GEMM tails (the 41st round of the
N = 1024 GEMMs, the partial last rounds of QKV / c_fc) and kernel boundaries run beside the other half's launches.
Per-row summation order does not depend on the split: the features must be bit-identical.  Interleaved A / B in one
process.  Run on the GPU box:

    python tools/bench_two_streams.py [frames] [repeats]
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eventclip_amd import _lib  # noqa: E402
from eventclip_amd import clip as eclip  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 2560
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cfg = eclip.arch_config('ViT-L/14')
m = eclip.CLIP(cfg, eclip.random_state_dict(cfg, seed=2), chunk=frames).cuda().eval()
pk = m._pack()
G = (cfg['image_size'] // cfg['patch']) ** 2
patches = (torch.randn(frames, G, pk['kpad'], device='cuda') * 0.5).half()
lib = _lib.lib()


def ws_for(n):
    need = lib.ec_vit_workspace_bytes(ctypes.byref(pk['vit']), n)
    return torch.empty((need,), dtype=torch.uint8, device='cuda')


def encode(p, out, ws, stream):
    rc = lib.ec_vit_encode(ctypes.byref(pk['vit']), _lib.ptr(p), p.shape[0], _lib.ptr(out), _lib.ptr(ws), ws.numel(),
                           p.shape[0], ctypes.c_void_p(stream.cuda_stream))
    _lib.check(rc, 'ec_vit_encode')


D = cfg['embed_dim']
half = frames // 2
ws_full, ws_a, ws_b = ws_for(frames), ws_for(half), ws_for(frames - half)
out_one = torch.empty(frames, D, device='cuda')
out_two = torch.empty(frames, D, device='cuda')
main = torch.cuda.current_stream()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def one_stream():
    encode(patches, out_one, ws_full, main)


def two_streams():
    fork = torch.cuda.Event()
    fork.record(main)
    s1.wait_event(fork)
    s2.wait_event(fork)
    encode(patches[:half], out_two[:half], ws_a, s1)
    encode(patches[half:], out_two[half:], ws_b, s2)
    j1, j2 = torch.cuda.Event(), torch.cuda.Event()
    j1.record(s1)
    j2.record(s2)
    main.wait_event(j1)
    main.wait_event(j2)


def halves_one_stream():
    encode(patches[:half], out_two[:half], ws_a, main)
    encode(patches[half:], out_two[half:], ws_b, main)


forms = {'one stream, one chunk (shipped)': one_stream, 'two halves on two streams': two_streams,
         'two halves, one stream (control)': halves_one_stream}
for fn in forms.values():
    fn()
torch.cuda.synchronize()
assert torch.equal(out_one, out_two), 'features differ between the split forms'
times = {k: [] for k in forms}
for _ in range(reps):
    for k, fn in forms.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        for _ in range(2):
            fn()
        e1.record(main)
        torch.cuda.synchronize()
        times[k].append(e0.elapsed_time(e1) / 2)
for k, t in times.items():
    t = sorted(t)
    print(f'{k:40s}: median {t[len(t) // 2]:8.2f} ms per {frames} frames  (min {t[0]:.2f}, max {t[-1]:.2f})', flush=True)
print('features bit-identical across the three forms')
