"""Thin Python wrappers over the low-level C-ABI kernels (unit tests, micro-benchmarks).

Tensors are torch CUDA tensors used as plain device buffers; everything runs on
torch's current stream.  No CPU fallback.
"""
import ctypes

from . import _lib


def dtype_code(t):
    import torch
    if t == torch.float16:
        return _lib.EC_F16
    if t == torch.bfloat16:
        return _lib.EC_BF16
    raise TypeError(f'16-bit dtype expected, got {t}')


def gemm(A, W, bias=None, epilogue='store16', out=None, variant=0, diag=None, resid=None, aux=None,
         splits=1, K=None, ws=None, row_stats=None, col_sums=None, row_stats_stride=0, row_sums=None,
         A_lo=None, W_lo=None, A_lo8=None, W8=None, A8=None, W_lo8=None, aux8=None):
    """out = epi(A[M,K] @ W[N,K]^T + bias).  epilogue: store16 | gelu16 | resid32 | store32 |
    gelu16_save (aux receives the pre-activation) | gelu_bwd16 (out = acc * QuickGELU'(aux)).
    resid32 accumulates into ``out`` (fp32) in place, or computes out = resid + ... when ``resid`` is given.
    splits > 1: A [M, splits * K], W [N, splits * K] -> out [splits, M, N] partial products (fp32).
    LayerNorm folded into the GEMM: resid_hl (``out`` = the hi plane in A's dtype, ``aux`` = the fp16 lo plane, both
    updated in place: (hi, lo) <- split(hi + lo + acc + bias)); store16_ln / gelu16_ln (``row_stats`` fp32 [M, 2] =
    (rstd, -rstd mean) of A's rows from ``row_stats``, ``col_sums`` fp32 [N] = row sums of W as rounded).
    Split-precision operands in one launch: ``A_lo`` / ``W_lo`` (the lo parts, same shapes and strides):
    out = epi(A_lo W^T + A W_lo^T + A W^T + bias); with them store16 / gelu16 take ``aux`` as the output's lo part.
    Lo products on the FP8 matrix path: ``A_lo8`` + ``W8`` in place of A_lo W^T, ``A8`` + ``W_lo8`` in place of A W_lo^T, each
    a pair (uint8 tensor [rows, 2 K] whose first K bytes per row are e4m3, exponent) as quantize_e4m3 returns them."""
    import torch
    _lib.require_gpu()
    epi = {'store16': _lib.EC_EPI_STORE16, 'gelu16': _lib.EC_EPI_GELU16,
           'resid32': _lib.EC_EPI_RESID32, 'store32': _lib.EC_EPI_STORE32,
           'gelu16_save': _lib.EC_EPI_GELU16_SAVE, 'gelu_bwd16': _lib.EC_EPI_GELU_BWD16,
           'resid_hl': _lib.EC_EPI_RESID_HL, 'store16_ln': _lib.EC_EPI_STORE16_LN, 'gelu16_ln': _lib.EC_EPI_GELU16_LN}[epilogue]
    M, KA = A.shape
    N = W.shape[0]
    if K is None:
        assert KA % splits == 0
        K = KA // splits
    assert W.shape[1] == KA and A.dtype == W.dtype and W.stride(1) == 1 and A.stride(1) == 1
    out16 = epi in (_lib.EC_EPI_STORE16, _lib.EC_EPI_GELU16, _lib.EC_EPI_GELU16_SAVE, _lib.EC_EPI_GELU_BWD16,
                    _lib.EC_EPI_RESID_HL, _lib.EC_EPI_STORE16_LN, _lib.EC_EPI_GELU16_LN)
    want = A.dtype if out16 else torch.float32
    shape = (M, N) if splits == 1 else (splits, M, N)
    if out is None:
        assert epilogue != 'resid32' or resid is not None, 'resid32 needs the fp32 residual tensor as out'
        out = torch.empty(shape, dtype=want, device=A.device)
    assert (epilogue != 'resid_hl' or aux is not None) and (not epilogue.endswith('_ln') or row_stats is not None)
    assert out.dtype == want and tuple(out.shape) == shape and out.stride(-1) == 1, \
        f'{epilogue} writes a {want} {shape} tensor, got {out.dtype} {tuple(out.shape)}'
    a = _lib.EcGemmArgs()
    a.M, a.N, a.K = M, N, K
    a.dtype, a.epilogue, a.variant = dtype_code(A.dtype), epi, variant
    a.A, a.lda = A.data_ptr(), A.stride(0)
    a.W, a.ldw = W.data_ptr(), W.stride(0)
    a.bias = bias.data_ptr() if bias is not None else None
    a.C, a.ldc = out.data_ptr(), out.stride(-2)
    a.diag = diag.data_ptr() if diag is not None else None    # EC_GEMM_DIAG builds only
    if resid is not None:
        assert resid.dtype == torch.float32 and tuple(resid.shape) == (M, N) and resid.stride(0) == out.stride(-2)
        a.resid = resid.data_ptr()
    if aux is not None:
        want_aux = torch.float16 if epilogue == 'resid_hl' else A.dtype          # the lo plane is always fp16
        assert aux.dtype == want_aux and tuple(aux.shape) == (M, N) and aux.stride(0) == out.stride(-2)
        a.aux = aux.data_ptr()
    if row_stats is not None:
        assert row_stats.dtype == torch.float32 and row_stats.is_contiguous() and col_sums.dtype == torch.float32
        a.row_stats, a.col_sums, a.row_stats_stride = row_stats.data_ptr(), col_sums.data_ptr(), row_stats_stride
    if row_sums is not None:      # resid_hl: (sum, sum of squares) of the new hi plane per 64-column group
        assert row_sums.dtype == torch.float32 and row_sums.is_contiguous() and row_sums.numel() == M * (N // 64) * 2
        a.row_sums = row_sums.data_ptr()
    if A_lo is not None:
        assert A_lo.dtype == A.dtype and A_lo.shape == A.shape and A_lo.stride() == A.stride()
        a.A_lo = A_lo.data_ptr()
    if W_lo is not None:
        assert W_lo.dtype == W.dtype and W_lo.shape == W.shape and W_lo.stride() == W.stride()
        a.W_lo = W_lo.data_ptr()
    for name, part, ref, ld in (('A_lo8', A_lo8, A, a.lda), ('W8', W8, W, a.ldw), ('A8', A8, A, a.lda), ('W_lo8', W_lo8, W, a.ldw)):
        if part is not None:
            t, e = part
            assert t.dtype == torch.uint8 and t.shape[0] == ref.shape[0] and t.stride(0) == 2 * ld and t.stride(1) == 1, name
            setattr(a, name, t.data_ptr())
            setattr(a, {'A_lo8': 'a_lo8_exp', 'W8': 'w8_exp', 'A8': 'a8_exp', 'W_lo8': 'w_lo8_exp'}[name], int(e))
    if aux8 is not None:          # gelu16 with e4m3 lo products: the lo output as e4m3 (uint8 [M, 2 N], exponent)
        t, e = aux8
        assert t.dtype == torch.uint8 and tuple(t.shape) == (M, 2 * N) and t.stride(0) == 2 * out.stride(-2) and aux is None
        a.aux, a.aux_e4m3, a.aux_exp = t.data_ptr(), 1, int(e)
    if splits > 1:
        a.splits, a.split_stride = splits, out.stride(0)
    if ws is not None:                       # fp32 scratch: an under-filled launch runs K-batched (low latency)
        assert ws.is_cuda and ws.is_contiguous()
        a.ws, a.ws_bytes = ws.data_ptr(), ws.numel() * ws.element_size()
    _lib.check(_lib.lib().ec_gemm(ctypes.byref(a), _lib.stream_ptr()), 'ec_gemm')
    return out


def quantize_e4m3(x, exp=None, pitch=None):
    """x [rows, K] (any float dtype, any device) -> (uint8 [rows, pitch or 2 K] on x's device, exp): the first K bytes of each
    row hold round_e4m3(x * 2^exp) (OCP e4m3fn, saturating at +-448), the layout of ec_gemm_args' e4m3 operands (the byte
    row pitch of the 16-bit operand).  exp=None: the largest |x| lands in [128, 256)."""
    import math
    import torch
    xf = x.detach().float()
    if exp is None:
        m = float(xf.abs().max())
        exp = 8 - int(math.ceil(math.log2(m))) if m > 0 else 0
    q = (xf * (2.0 ** exp)).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)
    rows, K = q.shape
    out = torch.zeros(rows, pitch or 2 * K, dtype=torch.uint8, device=x.device)
    out[:, :K] = q
    return out, exp


def dequantize_e4m3(t, exp, K):
    """the fp32 values an e4m3 operand of quantize_e4m3 / ec_layernorm_hl8 stands for"""
    import torch
    return t[:, :K].contiguous().view(torch.float8_e4m3fn).float() * (2.0 ** -exp)


def gemm_rows(A, W, splits=1, out=None):
    """out[s] = A[rows_s, M]^T @ W[rows_s, N] (fp32) over ``splits`` consecutive row ranges of the two row-major
    16-bit operands (ec_gemm_args.transposed: a weight gradient dY^T X without transposed copies).  Returns
    [M, N] (splits == 1) or the [splits, M, N] partial products."""
    import torch
    _lib.require_gpu()
    rows, M = A.shape
    N = W.shape[1]
    assert W.shape[0] == rows and A.dtype == W.dtype and A.stride(1) == 1 and W.stride(1) == 1
    K = ((rows + splits - 1) // splits + 63) // 64 * 64
    shape = (M, N) if splits == 1 else (splits, M, N)
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=A.device)
    assert out.dtype == torch.float32 and tuple(out.shape) == shape and out.stride(-1) == 1
    a = _lib.EcGemmArgs()
    a.M, a.N, a.K = M, N, K
    a.dtype, a.epilogue, a.variant = dtype_code(A.dtype), _lib.EC_EPI_STORE32, 0
    a.A, a.lda = A.data_ptr(), A.stride(0)
    a.W, a.ldw = W.data_ptr(), W.stride(0)
    a.C, a.ldc = out.data_ptr(), out.stride(-2)
    a.transposed, a.k_rows = 1, rows
    if splits > 1:
        a.splits, a.split_stride = splits, out.stride(0)
    _lib.check(_lib.lib().ec_gemm(ctypes.byref(a), _lib.stream_ptr()), 'ec_gemm')
    return out


def layernorm_hl(x_hi, x_lo, gamma, beta, eps=1e-5):
    """LayerNorm of the rows of hi + lo (x_lo: the fp16 lo plane) -> (hi, lo) parts in x_hi's dtype (ec_layernorm_hl)."""
    import torch
    _lib.require_gpu()
    rows, width = x_hi.shape
    assert x_lo.dtype == torch.float16 and x_lo.shape == x_hi.shape and x_lo.stride() == x_hi.stride()
    out = torch.empty((2, rows, width), dtype=x_hi.dtype, device=x_hi.device)
    _lib.check(_lib.lib().ec_layernorm_hl(x_hi.data_ptr(), x_lo.data_ptr(), x_hi.stride(0), gamma.data_ptr(), beta.data_ptr(),
                                          rows, width, float(eps), out[0].data_ptr(), out[1].data_ptr(), width,
                                          dtype_code(x_hi.dtype), _lib.stream_ptr()), 'ec_layernorm_hl')
    return out[0], out[1]


def row_stats(x16, eps=1e-5):
    """(rstd, -rstd * mean) of the rows of a 16-bit [rows, width] tensor -> fp32 [rows, 2] (ec_row_stats)."""
    import torch
    _lib.require_gpu()
    rows, width = x16.shape
    assert x16.stride(1) == 1
    out = torch.zeros((rows + 1, 2), dtype=torch.float32, device=x16.device)[:rows]   # readable to an even row count
    _lib.check(_lib.lib().ec_row_stats(x16.data_ptr(), x16.stride(0), rows, width, float(eps), out.data_ptr(),
                                       dtype_code(x16.dtype), _lib.stream_ptr()), 'ec_row_stats')
    return out


def row_stats_merge(row_sums, width, eps=1e-5):
    """[rows, width / 64, 2] partial sums of resid_hl -> fp32 [rows, 2] (rstd, -rstd * mean) (ec_row_stats_merge)."""
    import torch
    rows, groups = row_sums.shape[0], row_sums.shape[1]
    out = torch.zeros((rows + 1, 2), dtype=torch.float32, device=row_sums.device)[:rows]   # readable to an even row count
    _lib.check(_lib.lib().ec_row_stats_merge(row_sums.data_ptr(), rows, groups, width, float(eps), out.data_ptr(),
                                             _lib.stream_ptr()), 'ec_row_stats_merge')
    return out
