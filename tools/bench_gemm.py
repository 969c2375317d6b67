"""Micro-benchmark of the MFMA GEMM variants at the ViT-L/14 block shapes (run on the GPU box).

    python tools/bench_gemm.py [--frames 256] [--iters 10]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eventclip_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=256)
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--variants', type=int, nargs='+', default=[1, 4, 5])
    ap.add_argument('--dtypes', nargs='+', default=['float16', 'bfloat16'])
    a = ap.parse_args()
    M = a.frames * 257
    shapes = [('qkv', 3072, 1024, 'store16'), ('out', 1024, 1024, 'resid32'),
              ('fc1', 4096, 1024, 'gelu16'), ('fc2', 1024, 4096, 'resid32')]
    for dt in a.dtypes:
        dtype = getattr(torch, dt)
        for name, N, K, epi in shapes:
            A = torch.randn(M, K, device='cuda').to(dtype)
            W = (torch.randn(N, K, device='cuda') / K ** 0.5).to(dtype)
            bias = torch.randn(N, device='cuda')
            out = torch.zeros(M, N, device='cuda',
                              dtype=torch.float32 if epi == 'resid32' else dtype)
            for v in a.variants:
                for _ in range(10):
                    ops.gemm(A, W, bias, epi, out=out, variant=v)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    ops.gemm(A, W, bias, epi, out=out, variant=v)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / a.iters
                tf = 2. * M * N * K / ms / 1e9
                print(f'{dt:9s} {name} M={M} N={N} K={K} variant={v}: {ms:8.3f} ms '
                      f'{tf:7.1f} TFLOP/s', flush=True)


if __name__ == '__main__':
    main()
