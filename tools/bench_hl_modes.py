"""A / B of the hi-lo residual epilogue's forms (diagnostic build, ec_gemm variants 30 .. 33 = epilogue_hl_buf MODE
0 .. 3: bit 0 two scratch buffers / pipelined transposes, bit 1 growing residual prefetch) on the tower's two residual
GEMMs, interleaved in one process; results must be bit-identical.  Run on the GPU box:

    python tools/bench_hl_modes.py [frames]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('EVENTCLIP_HIP_LIB', os.path.join(ROOT, 'eventclip_amd', 'libeventclip_hip_diag.so'))
from eventclip_amd import ops  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 2560
VARIANTS = (0, 30, 31, 32, 33, 13)
M = frames * 257
for name, N, K in (('out_proj', 1024, 1024), ('c_proj', 1024, 4096)):
    g = torch.Generator(device='cuda').manual_seed(N + K)
    A = torch.randn(M, K, device='cuda', generator=g).half()
    W = (torch.randn(N, K, device='cuda', generator=g) / K ** 0.5).half()
    bias = torch.randn(N, device='cuda', generator=g)
    hi0 = torch.randn(M, N, device='cuda', generator=g).half()
    lo0 = (torch.randn(M, N, device='cuda', generator=g) * 1e-4).half()
    rs = torch.zeros(M, N // 64, 2, device='cuda')
    ref = None
    res = {}
    for v in VARIANTS:
        hi, lo = hi0.clone(), lo0.clone()
        rs.zero_()
        ops.gemm(A, W, bias, 'resid_hl', out=hi, aux=lo, row_sums=rs, variant=v)
        torch.cuda.synchronize()
        cur = (hi.clone(), lo.clone(), rs.clone())
        if ref is None:
            ref = cur
        assert all(torch.equal(a, b) for a, b in zip(cur, ref)), f'variant {v} differs'
    hi, lo = hi0.clone(), lo0.clone()
    times = {v: [] for v in VARIANTS}
    for _ in range(5):
        for v in times:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(6):
                ops.gemm(A, W, bias, 'resid_hl', out=hi, aux=lo, row_sums=rs, variant=v)
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / 6)
    for v, t in times.items():
        t = sorted(t)
        tag = 'default (product)' if v == 0 else 'two 4-wave workgroups per CU, 128 x 256 x 32 tiles' if v == 13 else f'MODE {v - 30}: ' + ('pipelined transposes' if (v - 30) & 1 else 'one scratch buffer') + \
            (', growing prefetch' if (v - 30) & 2 else ', prefetch 3 ahead')
        print(f'{name:9s} N={N} K={K}  {tag:55s}: median {t[2]:.3f} ms = {2.0 * M * N * K / t[2] / 1e9:6.0f} TFLOP/s', flush=True)
    del A, W, hi0, lo0, hi, lo
