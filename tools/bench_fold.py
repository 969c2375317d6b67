"""LayerNorm folded into the GEMMs around it (EC_EPI_RESID_HL / EC_EPI_STORE16_LN / EC_EPI_GELU16_LN + ec_row_stats)
against the plain chain (EC_EPI_RESID32 -> ec_layernorm -> EC_EPI_STORE16 / EC_EPI_GELU16): numerics against fp32
torch on a small shape, then the time of one block's GEMM + LayerNorm chain at the bench's M = 657 920, W = 1024,
both chains interleaved in one process.

    python tools/bench_fold.py [--frames 2560] [--rounds 3]
"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from eventclip_amd import _lib  # noqa: E402


def gemm(M, N, K, A, W, bias, C, epi, ldc=0, aux=None, row_stats=None, col_sums=None, dtype=_lib.EC_F16, row_sums=None):
    a = _lib.EcGemmArgs()
    a.M, a.N, a.K, a.dtype, a.epilogue, a.variant = M, N, K, dtype, epi, 0
    a.A, a.W, a.bias, a.C, a.ldc = _lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(C), ldc
    a.aux = _lib.ptr(aux) if aux is not None else None
    a.row_stats = _lib.ptr(row_stats) if row_stats is not None else None
    a.col_sums = _lib.ptr(col_sums) if col_sums is not None else None
    a.row_sums = _lib.ptr(row_sums) if row_sums is not None else None
    _lib.check(_lib.lib().ec_gemm(ctypes.byref(a), _lib.stream_ptr()), 'ec_gemm')


def fold_weights(Wt, b, gamma, beta):
    """(W' 16-bit, colsum(W' as rounded), b + W beta) for LN(x) W^T + b."""
    Wp = (Wt.float() * gamma[None, :]).half()
    return Wp, Wp.float().sum(1).contiguous(), (b + Wt.float() @ beta).contiguous()


def check():
    torch.manual_seed(0)
    M, Wd = 777, 256
    x = torch.randn(M, Wd, device='cuda') * 3 + 0.5
    hi = x.half()
    lo = (x - hi.float()).half()
    # producer: (hi, lo) <- split(hi + lo + A W^T + b)
    A = (torch.randn(M, 128, device='cuda')).half()
    Wo = (torch.randn(Wd, 128, device='cuda') * 0.1).half()
    bo = torch.randn(Wd, device='cuda')
    want = hi.float() + lo.float() + A.float() @ Wo.float().T + bo
    gemm(M, Wd, 128, A, Wo, bo, hi, _lib.EC_EPI_RESID_HL, aux=lo)
    got = hi.float() + lo.float()
    print('RESID_HL max rel err', float((got - want).abs().max() / want.abs().max()),
          ' hi == round(x):', bool(torch.equal(hi, want.half()) or (hi.float() - want).abs().max() < 2e-3 * want.abs().max()))
    # statistics out of the producer's epilogue
    x2 = torch.randn(M, Wd, device='cuda') * 3 + 0.5
    hi2, lo2 = x2.half(), (x2 - x2.half().float()).half()
    sums = torch.empty(M, Wd // 64, 2, device='cuda')
    gemm(M, Wd, 128, A, Wo, bo, hi2, _lib.EC_EPI_RESID_HL, aux=lo2, row_sums=sums)
    st2 = torch.empty(M + 1, 2, device="cuda")[:M]
    _lib.check(_lib.lib().ec_row_stats_merge(_lib.ptr(sums), M, Wd // 64, Wd, 1e-5, _lib.ptr(st2), _lib.stream_ptr()))
    h2 = hi2.float()
    r2 = (h2.var(1, unbiased=False) + 1e-5).rsqrt()
    print('fused sums: rstd rel err', float(((st2[:, 0] - r2) / r2).abs().max()), ' -rstd*mean abs err', float((st2[:, 1] + r2 * h2.mean(1)).abs().max()))
    # statistics of the hi plane
    stats = torch.empty(M + 1, 2, device="cuda")[:M]
    _lib.check(_lib.lib().ec_row_stats(_lib.ptr(hi), Wd, M, Wd, 1e-5, _lib.ptr(stats), _lib.EC_F16, _lib.stream_ptr()))
    h32 = hi.float()
    mean, var = h32.mean(1), h32.var(1, unbiased=False)
    rstd = (var + 1e-5).rsqrt()
    print('row_stats err', float((stats[:, 0] - rstd).abs().max() / rstd.abs().max()),
          float((stats[:, 1] + rstd * mean).abs().max()))
    # consumer: LN(hi) W^T + b
    gamma, beta = 1 + 0.2 * torch.randn(Wd, device='cuda'), 0.3 * torch.randn(Wd, device='cuda')
    Wq = (torch.randn(3 * Wd, Wd, device='cuda') * Wd ** -0.5)
    bq = torch.randn(3 * Wd, device='cuda') * 0.1
    Wp, cs, bf = fold_weights(Wq, bq, gamma, beta)
    ref = torch.nn.functional.layer_norm(h32, (Wd,), gamma, beta, 1e-5) @ Wq.T + bq
    for epi, f in ((_lib.EC_EPI_STORE16_LN, lambda t: t), (_lib.EC_EPI_GELU16_LN, lambda t: t * torch.sigmoid(1.702 * t))):
        out = torch.empty(M, 3 * Wd, dtype=torch.float16, device='cuda')
        gemm(M, 3 * Wd, Wd, hi, Wp, bf, out, epi, row_stats=stats, col_sums=cs)
        w = f(ref)
        print('consumer epi', epi, 'max err / max', float((out.float() - w).abs().max() / w.abs().max()))
        # the plain chain for comparison: LN -> 16 bit -> GEMM
        h = torch.nn.functional.layer_norm(h32, (Wd,), gamma, beta, 1e-5).half()
        plain = torch.empty_like(out)
        gemm(M, 3 * Wd, Wd, h, Wq.half(), bq, plain, _lib.EC_EPI_STORE16 if epi == _lib.EC_EPI_STORE16_LN else _lib.EC_EPI_GELU16)
        print('   plain chain        max err / max', float((plain.float() - w).abs().max() / w.abs().max()))


def bench(frames, rounds):
    S, Wd = 257, 1024
    M = frames * S
    dev = 'cuda'
    f16 = torch.float16
    x32 = torch.randn(M, Wd, device=dev)
    hi, lo = x32.half(), (x32 - x32.half().float()).half()
    h = torch.empty(M, Wd, dtype=f16, device=dev)           # LN output / attention output
    att = torch.randn(M, Wd, device=dev).half()
    qkv = torch.empty(M, 3 * Wd, dtype=f16, device=dev)
    mlp = torch.empty(M, 4 * Wd, dtype=f16, device=dev)
    stats = torch.empty(M + 1, 2, device=dev)[:M]
    g1, b1 = torch.ones(Wd, device=dev), torch.zeros(Wd, device=dev)
    mk = lambda n, k: (torch.randn(n, k, device=dev) * k ** -0.5).half()     # noqa: E731
    Wqkv, Wout, Wfc1, Wfc2 = mk(3 * Wd, Wd), mk(Wd, Wd), mk(4 * Wd, Wd), mk(Wd, 4 * Wd)
    bq, bo, bf1, bf2 = (torch.zeros(n, device=dev) for n in (3 * Wd, Wd, 4 * Wd, Wd))
    csq, cs1 = Wqkv.float().sum(1).contiguous(), Wfc1.float().sum(1).contiguous()
    L = _lib.lib()

    def ln(src):
        _lib.check(L.ec_layernorm(_lib.ptr(src), Wd, None, _lib.ptr(g1), _lib.ptr(b1), M, Wd, 1e-5, _lib.ptr(h), Wd,
                                  _lib.EC_F16, _lib.stream_ptr()))

    def rs():
        _lib.check(L.ec_row_stats(_lib.ptr(hi), Wd, M, Wd, 1e-5, _lib.ptr(stats), _lib.EC_F16, _lib.stream_ptr()))

    def plain():
        ln(x32)
        gemm(M, 3 * Wd, Wd, h, Wqkv, bq, qkv, _lib.EC_EPI_STORE16)
        gemm(M, Wd, Wd, att, Wout, bo, x32, _lib.EC_EPI_RESID32)
        ln(x32)
        gemm(M, 4 * Wd, Wd, h, Wfc1, bf1, mlp, _lib.EC_EPI_GELU16)
        gemm(M, Wd, 4 * Wd, mlp, Wfc2, bf2, x32, _lib.EC_EPI_RESID32)

    def folded():
        rs()
        gemm(M, 3 * Wd, Wd, hi, Wqkv, bq, qkv, _lib.EC_EPI_STORE16_LN, row_stats=stats, col_sums=csq)
        gemm(M, Wd, Wd, att, Wout, bo, hi, _lib.EC_EPI_RESID_HL, aux=lo)
        rs()
        gemm(M, 4 * Wd, Wd, hi, Wfc1, bf1, mlp, _lib.EC_EPI_GELU16_LN, row_stats=stats, col_sums=cs1)
        gemm(M, Wd, 4 * Wd, mlp, Wfc2, bf2, hi, _lib.EC_EPI_RESID_HL, aux=lo)

    sums = torch.empty(M, Wd // 64, 2, device=dev)

    def merge():
        _lib.check(L.ec_row_stats_merge(_lib.ptr(sums), M, Wd // 64, Wd, 1e-5, _lib.ptr(stats), _lib.stream_ptr()))

    def folded_sums():      # the statistics come out of the residual GEMMs' epilogues (previous block's c_proj feeds ln_1)
        merge()
        gemm(M, 3 * Wd, Wd, hi, Wqkv, bq, qkv, _lib.EC_EPI_STORE16_LN, row_stats=stats, col_sums=csq)
        gemm(M, Wd, Wd, att, Wout, bo, hi, _lib.EC_EPI_RESID_HL, aux=lo, row_sums=sums)
        merge()
        gemm(M, 4 * Wd, Wd, hi, Wfc1, bf1, mlp, _lib.EC_EPI_GELU16_LN, row_stats=stats, col_sums=cs1)
        gemm(M, Wd, 4 * Wd, mlp, Wfc2, bf2, hi, _lib.EC_EPI_RESID_HL, aux=lo, row_sums=sums)

    res = {'plain': [], 'folded': [], 'folded_sums': []}
    for fn in (plain, folded, folded_sums):
        for _ in range(2):
            fn()
    torch.cuda.synchronize()
    for _ in range(rounds):
        for name, fn in (('plain', plain), ('folded', folded), ('folded_sums', folded_sums)):
            _lib.profile_begin()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                fn()
            e1.record()
            torch.cuda.synchronize()
            prof = _lib.profile_end()
            res[name].append((e0.elapsed_time(e1) / 4, {e['name']: round(e['total_ms'] / 4, 3) for e in prof}))
    for name, rows in res.items():
        best = min(rows, key=lambda r: r[0])
        print(f'{name:7s} ms per block (attention excluded): ' + ' '.join(f'{r[0]:.3f}' for r in rows), ' best breakdown', best[1], flush=True)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=2560)
    ap.add_argument('--rounds', type=int, default=3)
    a = ap.parse_args()
    check()
    bench(a.frames, a.rounds)
