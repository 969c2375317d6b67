"""Race screen for the default GEMM (persistent, staggered, LDS-DMA, LDS-transposed epilogues): every shape /
epilogue is run many times on the same inputs under memory load and must reproduce its first result bit for
bit, which in turn is checked against torch.  Also the modes added in round 2: transposed operands with row
batches (weight gradients), the K-batched low-latency path with its fixup kernel, the QuickGELU pair and gradient
epilogues.  Run on the GPU box: python tools/race_screen.py [repeats]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eventclip_amd import ops  # noqa: E402

_args = [a for a in sys.argv[1:] if not a.startswith('--')]
reps = int(_args[0]) if _args else 60
ONLY_FP8 = '--only-fp8' in sys.argv        # round 6: the new kernel forms alone (the others: profiles/r5_race_screen.txt)
shapes = [(65792, 3072, 1024), (65792, 1024, 1024), (65792, 4096, 1024), (65792, 1024, 4096), (70001, 768, 640),
          (33333, 1024, 64), (257 * 300, 512, 512), (9999, 1536, 2048), (300000, 256, 128)]
bad = 0
if ONLY_FP8:
    shapes = []
noise = torch.empty(64 << 20, device='cuda')          # a copy kernel between launches perturbs L2 / HBM timing
for (M, N, K) in shapes:
    g = torch.Generator(device='cuda').manual_seed(M + N + K)
    A = torch.randn(M, K, device='cuda', generator=g).half()
    W = (torch.randn(N, K, device='cuda', generator=g) / K ** 0.5).half()
    bias = torch.randn(N, device='cuda', generator=g)
    resid = torch.randn(M, N, device='cuda', generator=g)
    want = A.float() @ W.float().t() + bias
    for epi in ('store16', 'gelu16', 'resid32', 'store32'):
        first = None
        for r in range(reps):
            if epi == 'resid32':
                out = resid.clone()
            else:   # a poisoned output buffer: a store the kernel misses cannot hide behind the previous repeat's value
                out = torch.full((M, N), float('nan'), dtype=torch.float16 if epi in ('store16', 'gelu16') else torch.float32,
                                 device='cuda')
            got = ops.gemm(A, W, bias, epi, out=out)
            if r % 3 == 0:
                noise.add_(1.0)
            if first is None:
                first = got.clone()
                ref = want * torch.sigmoid(1.702 * want) if epi == 'gelu16' else (want + resid if epi == 'resid32' else want)
                tol = 2e-3 if epi in ('store16', 'gelu16') else 1e-4
                err = float((got.float() - ref).abs().max() / ref.abs().max())
                if not err <= tol:
                    bad += 1
                    print('MISMATCH vs torch', (M, N, K), epi, err)
            elif not torch.equal(got, first):
                bad += 1
                n = int((got != first).sum())
                print('NONDETERMINISTIC', (M, N, K), epi, 'run', r, n, 'elements differ', flush=True)
                break
    print('ok', (M, N, K), flush=True)


def repeat(label, fn, check):
    """fn() -> tensor (or tuple of tensors); the first result is checked, the others must equal it bit for bit."""
    global bad
    first = None
    for r in range(reps):
        got = fn()
        got = got if isinstance(got, tuple) else (got,)
        if r % 3 == 0:
            noise.add_(1.0)
        if first is None:
            first = tuple(g.clone() for g in got)
            err = check(*first)
            if err is not None:
                bad += 1
                print('MISMATCH vs torch', label, err)
        elif not all(torch.equal(a, b) for a, b in zip(got, first)):
            bad += 1
            print('NONDETERMINISTIC', label, 'run', r, flush=True)
            return
    print('ok', label, flush=True)


# transposed operands: C = A^T W over the rows, in row batches (ragged last K tile)
for (rows, M, N, splits) in () if ONLY_FP8 else ((16448, 1024, 3072, 5), (16448, 1024, 1024, 16), (16448, 4096, 1024, 4), (70001, 768, 768, 8),
                             (3001, 256, 512, 2), (16448, 1024, 4096, 4)):
    g = torch.Generator(device='cuda').manual_seed(rows + M + N)
    A = torch.randn(rows, M, device='cuda', generator=g).half()
    W = (torch.randn(rows, N, device='cuda', generator=g) / rows ** 0.5).half()
    want = A.float().t() @ W.float()

    def check(out, want=want, splits=splits):
        tot = out if splits == 1 else out.sum(0)
        e = float((tot - want).abs().max() / want.abs().max())
        return e if e > 2e-4 else None
    repeat(('rows', rows, M, N, splits), lambda: ops.gemm_rows(A, W, splits), check)

# a few frames: K-batched launches + fixup (ec_gemm_args.ws), every epilogue the fixup implements
ws = torch.empty(80 << 20, dtype=torch.uint8, device='cuda')
for (M, N, K) in () if ONLY_FP8 else ((257, 1024, 1024), (257, 3072, 1024), (514, 1024, 4096), (50, 768, 3072), (2570, 4096, 1024)):
    g = torch.Generator(device='cuda').manual_seed(M * 7 + N + K)
    A = torch.randn(M, K, device='cuda', generator=g).half()
    W = (torch.randn(N, K, device='cuda', generator=g) / K ** 0.5).half()
    bias = torch.randn(N, device='cuda', generator=g)
    resid = torch.randn(M, N, device='cuda', generator=g)
    want = A.float() @ W.float().t() + bias
    for epi in ('store16', 'gelu16', 'resid32', 'store32', 'gelu16_save'):
        def run(epi=epi):
            if epi == 'resid32':
                return ops.gemm(A, W, bias, epi, resid=resid, ws=ws)
            if epi == 'gelu16_save':
                aux = torch.empty((M, N), dtype=A.dtype, device='cuda')
                return ops.gemm(A, W, bias, epi, aux=aux, ws=ws), aux
            return ops.gemm(A, W, bias, epi, ws=ws)

        def check(out, *rest, epi=epi):
            ref = want * torch.sigmoid(1.702 * want) if epi.startswith('gelu16') else (want + resid if epi == 'resid32' else want)
            tol = 2e-3 if epi in ('store16', 'gelu16', 'gelu16_save') else 1e-4
            e = float((out.float() - ref).abs().max() / ref.abs().max())
            return e if e > tol else None
        repeat(('ws', M, N, K, epi), run, check)

# LayerNorm folded into the GEMMs (round 3): hi / lo residual planes, row statistics, LN-finishing epilogues
for (M, N, K) in () if ONLY_FP8 else ((65792, 1024, 1024), (65792, 1024, 4096), (70001, 768, 640), (257, 1024, 1024)):
    g = torch.Generator(device='cuda').manual_seed(M + N * 5 + K)
    A = torch.randn(M, K, device='cuda', generator=g).half()
    W = (torch.randn(N, K, device='cuda', generator=g) / K ** 0.5).half()
    bias = torch.randn(N, device='cuda', generator=g)
    x = torch.randn(M, N, device='cuda', generator=g) * 3
    hi0, lo0 = x.half(), (x - x.half().float()).half()
    want = hi0.float() + lo0.float() + A.float() @ W.float().t() + bias

    def run_hl():
        hi, lo = hi0.clone(), lo0.clone()
        ops.gemm(A, W, bias, 'resid_hl', out=hi, aux=lo)
        return hi, lo

    def check_hl(hi, lo):
        e = float((hi.float() + lo.float() - want).abs().max() / want.abs().max())
        return e if e > 2e-6 else None
    repeat(('resid_hl', M, N, K), run_hl, check_hl)
for (M, N, K) in () if ONLY_FP8 else ((65792, 3072, 1024), (65792, 4096, 1024), (70001, 768, 640), (514, 3072, 1024)):
    g = torch.Generator(device='cuda').manual_seed(M + N * 3 + K * 7)
    X = (torch.randn(M, K, device='cuda', generator=g) * 2 + 0.3).half()
    gamma, beta = 1 + 0.2 * torch.randn(K, device='cuda', generator=g), 0.3 * torch.randn(K, device='cuda', generator=g)
    Wt = torch.randn(N, K, device='cuda', generator=g) / K ** 0.5
    bias = torch.randn(N, device='cuda', generator=g) * 0.1
    Wp = (Wt * gamma[None, :]).half()
    cs, bf = Wp.float().sum(1).contiguous(), (bias + Wt @ beta).contiguous()
    ref = torch.nn.functional.layer_norm(X.float(), (K,), gamma, beta, 1e-5) @ Wt.t() + bias
    for epi in ('store16_ln', 'gelu16_ln'):
        def run_ln(epi=epi):
            st = ops.row_stats(X)
            out = torch.full((M, N), float('nan'), dtype=torch.float16, device='cuda')
            return ops.gemm(X, Wp, bf, epi, out=out, row_stats=st, col_sums=cs), st

        def check_ln(out, st, epi=epi):
            w = ref * torch.sigmoid(1.702 * ref) if epi == 'gelu16_ln' else ref
            e = float((out.float() - w).abs().max() / w.abs().max())
            return e if not e <= 2e-3 else None
        repeat((epi, M, N, K), run_ln, check_ln)

# split-precision products in one launch (round 5): segments a_lo . w / a . w_lo / a . w, every epilogue the tower runs on
# them, the 16-bit ones with and without the lo output
for (M, N, K) in () if ONLY_FP8 else ((65792, 3072, 1024), (65792, 1024, 4096), (70001, 768, 640), (257, 1024, 1024)):
    g = torch.Generator(device='cuda').manual_seed(M * 3 + N + K * 11)
    a = torch.randn(M, K, device='cuda', generator=g)
    w = torch.randn(N, K, device='cuda', generator=g) / K ** 0.5
    pa, pw = torch.empty((2, M, K), dtype=torch.float16, device='cuda'), torch.empty((2, N, K), dtype=torch.float16, device='cuda')
    pa[0], pw[0] = a.half(), w.half()
    pa[1], pw[1] = (a - pa[0].float()).half(), (w - pw[0].float()).half()
    bias = torch.randn(N, device='cuda', generator=g)
    x = torch.randn(M, N, device='cuda', generator=g) * 3
    hi0, lo0 = x.half(), (x - x.half().float()).half()
    full = (pa[0].double() + pa[1].double()) @ (pw[0].double() + pw[1].double()).t() + bias.double()
    for epi, with_lo in (('store32', False), ('store16', False), ('store16', True), ('gelu16', False), ('gelu16', True), ('resid_hl', False)):
        def run_seg(epi=epi, with_lo=with_lo):
            if epi == 'resid_hl':
                hi, lo = hi0.clone(), lo0.clone()
                ops.gemm(pa[0], pw[0], bias, epi, out=hi, aux=lo, A_lo=pa[1], W_lo=pw[1])
                return hi, lo
            out = torch.full((M, N), float('nan'), dtype=torch.float32 if epi == 'store32' else torch.float16, device='cuda')
            if with_lo:
                aux = torch.full((M, N), float('nan'), dtype=torch.float16, device='cuda')
                return ops.gemm(pa[0], pw[0], bias, epi, out=out, aux=aux, A_lo=pa[1], W_lo=pw[1]), aux
            return ops.gemm(pa[0], pw[0], bias, epi, out=out, A_lo=pa[1], W_lo=pw[1])

        def check_seg(out, *rest, epi=epi, with_lo=with_lo):
            ref = full + x.double() if epi == 'resid_hl' else (full * torch.sigmoid(1.702 * full) if epi == 'gelu16' else full)
            got = out.double() + (rest[0].double() if rest else 0)
            tol = 2e-5 if (rest or epi == 'store32') else 2e-3
            e = float((got - ref).abs().max() / ref.abs().max())
            return e if not e <= tol else None
        repeat(('segments', epi, 'lo out' if with_lo else '', M, N, K), run_seg, check_seg)

# lo products on the FP8 matrix path (round 6): e4m3 segments in front of the 16-bit ones, every epilogue the tolerance mode runs
# on them (STORE16 with the 16-bit lo output, GELU16 with the e4m3 lo output, RESID_HL), one and two e4m3 segments, + a 16-bit W_lo
for (M, N, K) in ((65792, 3072, 1024), (65792, 1024, 4096), (70001, 768, 768), (257, 1024, 1024)):
    g = torch.Generator(device='cuda').manual_seed(M * 3 + N + K * 13)
    a = torch.randn(M, K, device='cuda', generator=g)
    w = torch.randn(N, K, device='cuda', generator=g) / K ** 0.5
    a_hi, w_hi = a.half(), w.half()
    w_lo = (w - w_hi.float()).half()
    A_lo8, W8 = ops.quantize_e4m3(a - a_hi.float(), exp=12), ops.quantize_e4m3(w_hi)
    A8, W_lo8 = ops.quantize_e4m3(a_hi, exp=0), ops.quantize_e4m3(w - w_hi.float())
    bias = torch.randn(N, device='cuda', generator=g)
    x = torch.randn(M, N, device='cuda', generator=g) * 3
    hi0, lo0 = x.half(), (x - x.half().float()).half()
    base = a_hi.double() @ w_hi.double().t() + bias.double() + ops.dequantize_e4m3(*A_lo8, K).double() @ ops.dequantize_e4m3(*W8, K).double().t()
    two = base + ops.dequantize_e4m3(*A8, K).double() @ ops.dequantize_e4m3(*W_lo8, K).double().t()
    mixed = base + a_hi.double() @ w_lo.double().t()
    del a, w
    for epi, parts, lo_out in (('store32', 'one', ''), ('store16', 'two', 'f16'), ('store16', 'one', ''), ('gelu16', 'two', 'e4m3'),
                               ('gelu16', 'one', 'e4m3'), ('gelu16', 'one', ''), ('resid_hl', 'one', ''), ('resid_hl', 'mixed', '')):
        kw = dict(A_lo8=A_lo8, W8=W8)
        if parts == 'two':
            kw.update(A8=A8, W_lo8=W_lo8)
        if parts == 'mixed':
            kw.update(W_lo=w_lo)
        full = two if parts == 'two' else mixed if parts == 'mixed' else base

        def run_f8(epi=epi, kw=kw, lo_out=lo_out):
            if epi == 'resid_hl':
                hi, lo = hi0.clone(), lo0.clone()
                ops.gemm(a_hi, w_hi, bias, epi, out=hi, aux=lo, **kw)
                return hi, lo
            out = torch.full((M, N), float('nan'), dtype=torch.float32 if epi == 'store32' else torch.float16, device='cuda')
            if lo_out == 'f16':
                aux = torch.full((M, N), float('nan'), dtype=torch.float16, device='cuda')
                return ops.gemm(a_hi, w_hi, bias, epi, out=out, aux=aux, **kw), aux
            if lo_out == 'e4m3':
                aux8 = torch.full((M, 2 * N), 0x7f, dtype=torch.uint8, device='cuda')
                return ops.gemm(a_hi, w_hi, bias, epi, out=out, aux8=(aux8, 12), **kw), aux8
            return ops.gemm(a_hi, w_hi, bias, epi, out=out, **kw)

        def check_f8(out, *rest, epi=epi, full=full, lo_out=lo_out):
            ref = full + x.double() if epi == 'resid_hl' else (full * torch.sigmoid(1.702 * full) if epi == 'gelu16' else full)
            got = out.double()
            if epi == 'resid_hl' or lo_out == 'f16':
                got = got + rest[0].double()
            if lo_out == 'e4m3':
                got = got + ops.dequantize_e4m3(rest[0], 12, N).double()
            tol = 2e-5 if (epi in ('resid_hl', 'store32') or lo_out == 'f16') else (2e-4 if lo_out == 'e4m3' else 2e-3)
            e = float((got - ref).abs().max() / ref.abs().max())
            return e if not e <= tol else None
        repeat(('e4m3 segments', epi, parts, lo_out, M, N, K), run_f8, check_f8)

print('race screen:', 'CLEAN' if bad == 0 else f'{bad} problems', f'({reps} repeats per case)')
sys.exit(1 if bad else 0)
