"""One GEMM shape, one variant, a few launches: target for rocprofv3 --pmc runs."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eventclip_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--variant', type=int, default=4)
ap.add_argument('--n', type=int, default=3072)
ap.add_argument('--k', type=int, default=1024)
ap.add_argument('--frames', type=int, default=256)
ap.add_argument('--epi', default='store16')
ap.add_argument('--iters', type=int, default=3)
a = ap.parse_args()
M = a.frames * 257
A = torch.randn(M, a.k, device='cuda').half()
W = (torch.randn(a.n, a.k, device='cuda') / a.k ** 0.5).half()
bias = torch.randn(a.n, device='cuda')
out = torch.zeros(M, a.n, device='cuda', dtype=torch.float32 if a.epi in ('resid32', 'store32') else torch.float16)
for _ in range(a.iters):
    ops.gemm(A, W, bias, a.epi, out=out, variant=a.variant)
torch.cuda.synchronize()
