"""A / B of where the 16-bit epilogues request the next tile's first K tile (diagnostic build): behind the first use of
bias / column sums / row statistics (shipped) against in front of it (variant 43: there hipcc's waits for those loads
also wait for the LDS-DMA requests).  One process, interleaved, bit-identical outputs.

    python tools/bench_epi16.py [frames]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('EVENTCLIP_HIP_LIB', os.path.join(ROOT, 'eventclip_amd', 'libeventclip_hip_diag.so'))
from eventclip_amd import ops  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 2560
M = frames * 257
for name, N, K, epi in (('QKV', 3072, 1024, 'store16_ln'), ('c_fc', 4096, 1024, 'gelu16_ln'), ('QKV plain', 3072, 1024, 'store16'),
                        ('c_fc plain', 4096, 1024, 'gelu16')):
    g = torch.Generator(device='cuda').manual_seed(N + K)
    A = torch.randn(M, K, device='cuda', generator=g).half()
    W = (torch.randn(N, K, device='cuda', generator=g) / K ** 0.5).half()
    bias = torch.randn(N, device='cuda', generator=g)
    kw = dict(row_stats=ops.row_stats(A), col_sums=W.float().sum(1).contiguous()) if epi.endswith('_ln') else {}
    outs = {}
    for v in (0, 43):
        out = torch.empty(M, N, device='cuda', dtype=torch.float16)
        ops.gemm(A, W, bias, epi, out=out, variant=v, **kw)
        outs[v] = out
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[43])
    times = {0: [], 43: []}
    for _ in range(9):
        for v in times:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                ops.gemm(A, W, bias, epi, out=outs[v], variant=v, **kw)
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / 4)
    for v, t in times.items():
        t = sorted(t)
        print(f'{name:11s} N={N} K={K} {epi:10s} {"requests behind the first use (shipped)" if v == 0 else "requests in front of it":40s}: '
              f'median {t[4]:.3f} ms = {2.0 * M * N * K / t[4] / 1e9:6.0f} TFLOP/s', flush=True)
    del A, W, outs
