"""Where does the hi-lo residual epilogue's time go?  (one bounded experiment.)

The dominant kernel class of the step is gemm_kernel<RESID32> = out_proj + c_proj with EC_EPI_RESID_HL: per output element
it reads the two residual planes (4 B), writes them back (4 B) and leaves (sum, sum of squares) per 64 columns.  This tool
times, interleaved in one process on the tower's two shapes (diagnostic build), the product kernel against forms of the SAME
kernel with one part of the epilogue removed (wrong results on purpose):

    default            the product epilogue (planes read, planes written, row sums)
    no row sums        ec_gemm_args.row_sums = NULL
    no residual loads  variant 34: the planes read as zero (no HBM read, same arithmetic and stores)
    no lo store        variant 35
    no stores          variant 36
    no loads, no stores  variant 37: the epilogue's arithmetic + LDS transposes alone
    store16            the plain 16-bit store epilogue on the same product (2 B per element, no read)

    python tools/bench_resid_split.py [frames] > profiles/r6_resid_split.txt
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
FORMS = [('default', dict(variant=0), True), ('planes touched one K tile ahead (variant 38)', dict(variant=38), True), ('no row sums', dict(variant=0), False), ('no residual loads', dict(variant=34), True),
         ('no lo store', dict(variant=35), True), ('no stores', dict(variant=36), True), ('no loads, no stores', dict(variant=37), True),
         ('no loads, no stores, no row sums', dict(variant=37), False), ('store16 (plain 16-bit store)', None, False)]
for name, N, K in (('out_proj', 1024, 1024), ('c_proj', 1024, 4096)):
    g = torch.Generator(device='cuda').manual_seed(N + K)
    A = torch.randn(M, K, device='cuda', generator=g).half()
    W = (torch.randn(N, K, device='cuda', generator=g) / K ** 0.5).half()
    bias = torch.randn(N, device='cuda', generator=g)
    hi = torch.randn(M, N, device='cuda', generator=g).half()
    lo = (torch.randn(M, N, device='cuda', generator=g) * 1e-4).half()
    out16 = torch.empty(M, N, device='cuda', dtype=torch.float16)
    rs = torch.zeros(M, N // 64, 2, device='cuda')

    def run(kw, stats):
        if kw is None:
            ops.gemm(A, W, bias, 'store16', out=out16)
        else:
            ops.gemm(A, W, bias, 'resid_hl', out=hi, aux=lo, row_sums=rs if stats else None, **kw)
    times = {f[0]: [] for f in FORMS}
    for f in FORMS:
        run(f[1], f[2])
    torch.cuda.synchronize()
    # (variant 38 must give the product kernel's bits)
    h0, l0 = hi.clone(), lo.clone()
    run(dict(variant=0), True)
    r0 = (hi.clone(), lo.clone(), rs.clone())
    hi.copy_(h0), lo.copy_(l0)
    run(dict(variant=38), True)
    assert torch.equal(hi, r0[0]) and torch.equal(lo, r0[1]) and torch.equal(rs, r0[2]), 'variant 38 differs'
    for _ in range(7):
        for tag, kw, stats in FORMS:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(6):
                run(kw, stats)
            e1.record()
            torch.cuda.synchronize()
            times[tag].append(e0.elapsed_time(e1) / 6)
        hi.normal_()          # the in-place planes grow by the product each launch: keep them finite
        lo.zero_()
    base = sorted(times['default'])[3]
    for tag, t in times.items():
        t = sorted(t)[3]
        print(f'{name:9s} M={M} N={N} K={K}  {tag:46s}: median {t:.3f} ms = {2.0 * M * N * K / t / 1e9:6.0f} TFLOP/s  ({t - base:+.3f} ms vs default)',
              flush=True)
    del A, W, hi, lo, out16, rs
