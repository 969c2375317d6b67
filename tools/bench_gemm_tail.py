"""What the 41st round of the N = 1024 GEMMs costs, and what a K-split tail would recover: the full launch against
[the first 2560 row tiles = 40 whole rounds] + [the last 10 row tiles as a K-batched launch with its fix-up]
(ec_gemm_args.ws, the low-latency mode's path), same operands, interleaved.  Measurement only (profiles/r3_gemm.md 5).

    python tools/bench_gemm_tail.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eventclip_amd import ops  # noqa: E402

M, MAIN = 2560 * 257, 2560 * 256
ws = torch.empty(80 << 20, dtype=torch.uint8, device='cuda')
for name, N, K, epi in (('out_proj', 1024, 1024, 'resid32'), ('c_proj', 1024, 4096, 'resid32'), ('c_proj', 1024, 4096, 'store16')):
    g = torch.Generator(device='cuda').manual_seed(N + K)
    A = torch.randn(M, K, device='cuda', generator=g).half()
    W = (torch.randn(N, K, device='cuda', generator=g) / K ** 0.5).half()
    bias = torch.randn(N, device='cuda', generator=g)
    out = torch.zeros(M, N, dtype=torch.float32 if epi == 'resid32' else torch.float16, device='cuda')
    fns = {'one launch (41 rounds)': lambda: ops.gemm(A, W, bias, epi, out=out),
           'rows of 40 whole rounds': lambda: ops.gemm(A[:MAIN], W, bias, epi, out=out[:MAIN]),
           'last 2560 rows, one pass': lambda: ops.gemm(A[MAIN:], W, bias, epi, out=out[MAIN:]),
           'last 2560 rows, K-batched + fix-up': lambda: ops.gemm(A[MAIN:], W, bias, epi, out=out[MAIN:], ws=ws)}
    times = {k: [] for k in fns}
    for fn in fns.values():
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    for _ in range(5):
        for k, fn in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) / 10)
    med = {k: sorted(t)[2] for k, t in times.items()}
    for k, t in med.items():
        print(f'{name:9s} K={K} {epi:8s} {k:36s}: {t * 1e3:8.1f} us', flush=True)
    print(f'          split launch would take {1e3 * (med["rows of 40 whole rounds"] + med["last 2560 rows, K-batched + fix-up"]):.1f} us against '
          f'{1e3 * med["one launch (41 rounds)"]:.1f}', flush=True)
