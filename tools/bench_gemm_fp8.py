"""The lo products of the split-operand blocks on the FP8 matrix path against the 16-bit form (round 6), on the tower's
shapes at the bench size, interleaved in one process:

    plain        A W^T                                  (the default block's product)
    f16 lo       A_lo W^T + A W^T                       (a split-operand block on a checkpoint stored in 16 bit)
    e4m3 lo      dq(A_lo8) dq(W8)^T + A W^T
    f16 lo x 2   A_lo W^T + A W_lo^T + A W^T            (fp32 weights)
    e4m3 lo x 2  dq(A_lo8) dq(W8)^T + dq(A8) dq(W_lo8)^T + A W^T

    python tools/bench_gemm_fp8.py [frames] > profiles/r6_gemm_fp8.txt
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from eventclip_amd import ops  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith('--')]
ZERO = '--zero-operands' in sys.argv      # all-zero operands: the clock the chip holds without the power cap biting
frames = int(args[0]) if args else 2560
M = frames * 257
for name, N, K, epi in (('QKV', 3072, 1024, 'store16'), ('c_fc', 4096, 1024, 'gelu16'), ('out_proj', 1024, 1024, 'resid_hl'),
                        ('c_proj', 1024, 4096, 'resid_hl')):
    g = torch.Generator(device='cuda').manual_seed(N + K)
    a = torch.randn(M, K, device='cuda', generator=g) * (0.0 if ZERO else 1.0)
    w = torch.randn(N, K, device='cuda', generator=g) / K ** 0.5 * (0.0 if ZERO else 1.0)
    bias = torch.randn(N, device='cuda', generator=g)
    a_hi = a.half()
    a_lo = (a - a_hi.float()).half()
    w_hi = w.half()
    w_lo = (w - w_hi.float()).half()
    A_lo8, W8 = ops.quantize_e4m3(a - a_hi.float(), exp=12), ops.quantize_e4m3(w_hi, exp=4)
    A8, W_lo8 = ops.quantize_e4m3(a_hi, exp=0), ops.quantize_e4m3(w - w_hi.float(), exp=16)
    del a, w
    hi = torch.randn(M, N, device='cuda', generator=g).half()
    lo = torch.zeros(M, N, device='cuda', dtype=torch.float16)
    forms = [('plain', {}), ('f16 lo', dict(A_lo=a_lo)), ('e4m3 lo', dict(A_lo8=A_lo8, W8=W8)),
             ('f16 lo x 2', dict(A_lo=a_lo, W_lo=w_lo)), ('e4m3 lo x 2', dict(A_lo8=A_lo8, W8=W8, A8=A8, W_lo8=W_lo8))]
    out16 = torch.empty(M, N, device='cuda', dtype=torch.float16)

    def run(kw):
        if epi == 'resid_hl':
            ops.gemm(a_hi, w_hi, bias, epi, out=hi, aux=lo, **kw)
        else:
            ops.gemm(a_hi, w_hi, bias, epi, out=out16, **kw)
    times = {f[0]: [] for f in forms}
    for f in forms:
        run(f[1])
    torch.cuda.synchronize()
    for _ in range(5):
        for tag, kw in forms:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                run(kw)
            e1.record()
            torch.cuda.synchronize()
            times[tag].append(e0.elapsed_time(e1) / 5)
        hi.normal_()
        lo.zero_()
    base = sorted(times['plain'])[2]
    for tag, t in times.items():
        t = sorted(t)[2]
        print(('zero operands  ' if ZERO else '') + f'{name:9s} M={M} N={N} K={K} {epi:9s} {tag:12s}: median {t:.3f} ms ({t / base:.2f} x plain, +{t - base:.3f} ms)', flush=True)
    del a_hi, a_lo, w_hi, w_lo, A_lo8, W8, A8, W_lo8, hi, lo, out16
