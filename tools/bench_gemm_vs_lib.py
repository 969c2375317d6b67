"""The tower's four GEMM shapes (M = 2560 frames x 257 tokens) through ec_gemm and through the vendor library
behind torch.matmul (hipBLASLt / rocBLAS on this image), interleaved in one process on the same operands.
A yardstick for DESIGN 3.1, not a product path: nothing in eventclip_amd/ calls the library.

    python tools/bench_gemm_vs_lib.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eventclip_amd import ops  # noqa: E402

M = 2560 * 257
for name, N, K in (('QKV', 3072, 1024), ('out_proj', 1024, 1024), ('c_fc', 4096, 1024), ('c_proj', 1024, 4096)):
    g = torch.Generator(device='cuda').manual_seed(N + K)
    A = torch.randn(M, K, device='cuda', generator=g).half()
    W = (torch.randn(N, K, device='cuda', generator=g) / K ** 0.5).half()
    bias = torch.randn(N, device='cuda', generator=g)
    out = torch.empty(M, N, dtype=torch.float16, device='cuda')
    Wt = W.t()
    fns = {'ec_gemm store16': lambda: ops.gemm(A, W, bias, 'store16', out=out),
           'torch.matmul (vendor library)': lambda: torch.matmul(A, Wt, out=out),
           'torch.addmm (library + bias)': lambda: torch.addmm(bias.half(), A, Wt, out=out)}
    times = {k: [] for k in fns}
    for k, fn in fns.items():
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    for _ in range(5):
        for k, fn in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) / 5)
    for k, t in times.items():
        t = sorted(t)
        print(f'{name:9s} M={M} N={N} K={K}  {k:32s}: median {t[2]:.3f} ms = {2.0 * M * N * K / t[2] / 1e9:6.0f} TFLOP/s', flush=True)
    del A, W, out
