"""The tower's four GEMM shapes (M = 2560 frames x 257 tokens) through ec_gemm and through the vendor library
behind torch.matmul (hipBLASLt / rocBLAS on this image), interleaved in one process on the same operands.
A yardstick for DESIGN 3.1, not a product path: nothing in eventclip_amd/ calls the library.

    python tools/bench_gemm_vs_lib.py                   the comparison on random operands
    python tools/bench_gemm_vs_lib.py --zero-operands   the ceiling experiment (round 5): every shape with the epilogue the
        tower runs on it, on RANDOM and on ZERO-filled operands, ec_gemm and the library, plus the in-kernel shader
        clock of ec_gemm on both -- s_memtime / s_memrealtime x 100 MHz stamped once at each workgroup's start and end
        (variant 19 of the diagnostic build, after >= 2 s of back-to-back launches: MI355X_MICROARCH.md, DVFS give-back
        6; median over the workgroups).  TF/s(zero) / TF/s(random) is what the power cap takes; the clock says why.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument('--zero-operands', action='store_true')
ap.add_argument('--frames', type=int, default=2560)
a = ap.parse_args()
if a.zero_operands:
    os.environ.setdefault('EVENTCLIP_HIP_LIB', os.path.join(ROOT, 'eventclip_amd', 'libeventclip_hip_diag.so'))
from eventclip_amd import ops  # noqa: E402

M = a.frames * 257
SHAPES = (('QKV', 3072, 1024, 'store16_ln'), ('out_proj', 1024, 1024, 'resid_hl'), ('c_fc', 4096, 1024, 'gelu16_ln'),
          ('c_proj', 1024, 4096, 'resid_hl'))


def timed(fn, reps=5, inner=5):
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / inner)
    return sorted(ts)[len(ts) // 2]


def plain():
    for name, N, K, _ in SHAPES:
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
                times[k].append(timed(fn, reps=1))
        for k, t in times.items():
            t = sorted(t)
            print(f'{name:9s} M={M} N={N} K={K}  {k:32s}: median {t[2]:.3f} ms = {2.0 * M * N * K / t[2] / 1e9:6.0f} TFLOP/s', flush=True)
        del A, W, out


def ceiling():
    print(f'# M = {M} rows; TFLOP/s = 2 M N K / time; clock = median over workgroups of d(s_memtime) / d(s_memrealtime) x 100 MHz')
    for name, N, K, epi in SHAPES:
        res = {}
        for fill in ('random', 'zero'):
            g = torch.Generator(device='cuda').manual_seed(N + K)
            A = torch.randn(M, K, device='cuda', generator=g).half()
            W = (torch.randn(N, K, device='cuda', generator=g) / K ** 0.5).half()
            bias = torch.randn(N, device='cuda', generator=g)
            if fill == 'zero':
                A.zero_(), W.zero_(), bias.zero_()
            kw = {}
            if epi == 'resid_hl':
                out = torch.randn(M, N, device='cuda', generator=g).half()
                kw = dict(aux=torch.zeros(M, N, device='cuda', dtype=torch.float16), row_sums=torch.zeros(M, N // 64, 2, device='cuda'))
                if fill == 'zero':
                    out.zero_()
            else:
                out = torch.empty(M, N, device='cuda', dtype=torch.float16)
                kw = dict(row_stats=ops.row_stats(A), col_sums=W.float().sum(1).contiguous())
            out_lib = torch.empty(M, N, device='cuda', dtype=torch.float16)
            Wt, bh = W.t(), bias.half()
            ec = lambda: ops.gemm(A, W, bias, epi, out=out, **kw)              # noqa: E731
            lib = lambda: torch.addmm(bh, A, Wt, out=out_lib)                  # noqa: E731
            # >= 2 s of back-to-back launches so the chip is at its sustained state, then the clock launch right behind
            t_end = time.time() + 2.2
            while time.time() < t_end:
                for _ in range(20):
                    ec()
                torch.cuda.synchronize()
            nwg = min(((M + 255) // 256) * ((N + 255) // 256), 256)
            dbg = torch.zeros(nwg * 4 * 2, device='cuda', dtype=torch.float32)
            for _ in range(10):
                ec()
            ops.gemm(A, W, bias, epi, out=out, variant=19, diag=dbg, **kw)
            torch.cuda.synchronize()
            r = dbg.cpu().numpy().view(np.uint64).reshape(nwg, 4).astype(np.float64)
            ghz = np.median((r[:, 2] - r[:, 0]) / (r[:, 3] - r[:, 1])) * 0.1
            t_ec, t_lib = [], []
            for _ in range(5):
                t_ec.append(timed(ec, reps=1))
                t_lib.append(timed(lib, reps=1))
            t_ec, t_lib = sorted(t_ec)[2], sorted(t_lib)[2]
            flop = 2.0 * M * N * K
            res[fill] = (flop / t_ec / 1e9, flop / t_lib / 1e9, ghz)
            print(f'{name:9s} N={N} K={K} {epi:10s} {fill:6s}: ec_gemm {t_ec:.3f} ms = {flop / t_ec / 1e9:5.0f} TFLOP/s at {ghz:.3f} GHz in-kernel'
                  f' ({flop / t_ec / 1e9 / (2500.0 * ghz / 2.4):.3f} of the MFMA rate at that clock); library addmm {t_lib:.3f} ms = {flop / t_lib / 1e9:5.0f} TFLOP/s',
                  flush=True)
            del A, W, out, out_lib, kw
        print(f'{name:9s} TF/s(zero) / TF/s(random): ec_gemm {res["zero"][0] / res["random"][0]:.3f}, library {res["zero"][1] / res["random"][1]:.3f};'
              f' clock(zero) / clock(random) {res["zero"][2] / res["random"][2]:.3f}', flush=True)


if a.zero_operands:
    ceiling()
else:
    plain()
