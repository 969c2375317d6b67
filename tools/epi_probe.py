"""Epilogue-cost probe: GEMM time against K at the out_proj footprint (run on the GPU box)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eventclip_amd import ops  # noqa: E402

M = 2560 * 257


def run(N, K, epi, v, iters=5):
    A = torch.randn(M, K, device='cuda').half()
    W = (torch.randn(N, K, device='cuda') / K ** 0.5).half()
    bias = torch.randn(N, device='cuda')
    out = torch.zeros(M, N, device='cuda', dtype=torch.float32 if epi.endswith('32') else torch.float16)
    for _ in range(3):
        ops.gemm(A, W, bias, epi, out=out, variant=v)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.gemm(A, W, bias, epi, out=out, variant=v)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f'N={N} K={K} {epi} v={v}: {ms:.3f} ms  {2. * M * N * K / ms / 1e9:.0f} TF', flush=True)


if __name__ == '__main__':
    which = sys.argv[1] if len(sys.argv) > 1 else 'all'
    if which in ('all', 'k'):
        for K in (64, 128, 256, 512, 1024, 2048):
            run(1024, K, 'resid32', 5)
    if which in ('all', 's'):
        for K in (64, 1024):
            run(1024, K, 'store32', 5)
            run(1024, K, 'store16', 5)
    if which in ('all', 'd'):
        for v in (6, 7):
            run(1024, 1024, 'resid32', v)
            run(1024, 1024, 'store16', v)
    if which in ('all', 'rmw'):
        x = torch.zeros(M, 1024, device='cuda')
        for _ in range(3):
            x.add_(1.0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            x.add_(1.0)
        e1.record()
        torch.cuda.synchronize()
        print('torch x += 1 (5.4 GB):', e0.elapsed_time(e1) / 5, 'ms')
