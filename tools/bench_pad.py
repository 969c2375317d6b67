"""Do power-of-two row strides cost the GEMM's staging DMA anything (L2 channel hot spots: a 256-row K tile at a 2-KiB row
stride)?  The tower's shapes with A / W rows padded by 0 / 64 / 128 / 192 elements, store16, one process, interleaved.

    python tools/bench_pad.py [frames]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from eventclip_amd import ops  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 2560
M = frames * 257
PADS = (0, 64, 128, 192)


def padded(rows, cols, pad, gen, scale=1.0):
    t = torch.empty(rows, cols + pad, device='cuda', dtype=torch.float16)
    t[:, :cols] = (torch.randn(rows, cols, device='cuda', generator=gen) * scale).half()
    return t[:, :cols]


for name, N, K in (('QKV', 3072, 1024), ('out_proj', 1024, 1024), ('c_fc', 4096, 1024), ('c_proj', 1024, 4096)):
    g = torch.Generator(device='cuda').manual_seed(N + K)
    bias = torch.randn(N, device='cuda', generator=g)
    ops_in = {}
    ref = None
    for pad in PADS:
        g2 = torch.Generator(device='cuda').manual_seed(N + K)
        A = padded(M, K, pad, g2)
        W = padded(N, K, pad, g2, K ** -0.5)
        out = torch.empty(M, N + pad, device='cuda', dtype=torch.float16)[:, :N]
        ops.gemm(A, W, bias, 'store16', out=out)
        ops_in[pad] = (A, W, out)
        if ref is None:
            ref = out.clone()
        assert torch.equal(out, ref), pad
    times = {p: [] for p in PADS}
    for _ in range(7):
        for p in PADS:
            A, W, out = ops_in[p]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                ops.gemm(A, W, bias, 'store16', out=out)
            e1.record()
            torch.cuda.synchronize()
            times[p].append(e0.elapsed_time(e1) / 4)
    print(f'{name:9s} N={N} K={K}: ' + '  '.join(f'pad {p}: {sorted(t)[3]:.3f} ms' for p, t in times.items()), flush=True)
    del ops_in
