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
ap.add_argument('--lo', default='none', choices=['none', 'f16', 'e4m3'],
                help='add the lo product A_lo W^T as 16-bit or as e4m3 operands (round 6: the split-operand blocks of the tolerance mode)')
a = ap.parse_args()
M = a.frames * 257
A = torch.randn(M, a.k, device='cuda').half()
W = (torch.randn(a.n, a.k, device='cuda') / a.k ** 0.5).half()
bias = torch.randn(a.n, device='cuda')
out = torch.zeros(M, a.n, device='cuda', dtype=torch.float32 if a.epi in ('resid32', 'store32') else torch.float16)
kw = {}
if a.epi == 'resid_hl':      # the tower's residual GEMMs: hi / lo planes + row sums
    kw = dict(aux=torch.zeros(M, a.n, device='cuda', dtype=torch.float16), row_sums=torch.zeros(M, a.n // 64, 2, device='cuda'))
elif a.epi.endswith('_ln'):  # ... its LayerNorm-finishing GEMMs
    kw = dict(row_stats=ops.row_stats(A), col_sums=W.float().sum(1).contiguous())
if a.lo != 'none':
    full = torch.randn(M, a.k, device='cuda')
    A = full.half()
    if a.lo == 'f16':
        kw['A_lo'] = (full - A.float()).half()
    else:
        kw['A_lo8'], kw['W8'] = ops.quantize_e4m3(full - A.float(), exp=12), ops.quantize_e4m3(W)
    del full
for _ in range(a.iters):
    ops.gemm(A, W, bias, a.epi, out=out, variant=a.variant, **kw)
torch.cuda.synchronize()
