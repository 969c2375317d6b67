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


def gemm(A, W, bias=None, epilogue='store16', out=None, variant=0, diag=None):
    """out = epi(A[M,K] @ W[N,K]^T + bias).  epilogue: store16 | gelu16 | resid32 | store32.
    resid32 accumulates into ``out`` (fp32) in place."""
    import torch
    _lib.require_gpu()
    epi = {'store16': _lib.EC_EPI_STORE16, 'gelu16': _lib.EC_EPI_GELU16,
           'resid32': _lib.EC_EPI_RESID32, 'store32': _lib.EC_EPI_STORE32}[epilogue]
    M, K = A.shape
    N = W.shape[0]
    assert W.shape[1] == K and A.dtype == W.dtype and W.is_contiguous() and A.stride(1) == 1
    if out is None:
        assert epilogue != 'resid32', 'resid32 needs the fp32 residual tensor as out'
        odt = A.dtype if epi in (_lib.EC_EPI_STORE16, _lib.EC_EPI_GELU16) else torch.float32
        out = torch.empty((M, N), dtype=odt, device=A.device)
    want = A.dtype if epi in (_lib.EC_EPI_STORE16, _lib.EC_EPI_GELU16) else torch.float32
    assert out.dtype == want and tuple(out.shape) == (M, N) and out.stride(1) == 1, \
        f'{epilogue} writes a {want} [{M}, {N}] tensor, got {out.dtype} {tuple(out.shape)}'
    a = _lib.EcGemmArgs()
    a.M, a.N, a.K = M, N, K
    a.dtype, a.epilogue, a.variant = dtype_code(A.dtype), epi, variant
    a.A, a.lda = A.data_ptr(), A.stride(0)
    a.W = W.data_ptr()
    a.bias = bias.data_ptr() if bias is not None else None
    a.C, a.ldc = out.data_ptr(), out.stride(0)
    a.diag = diag.data_ptr() if diag is not None else None    # EC_GEMM_DIAG builds only
    _lib.check(_lib.lib().ec_gemm(ctypes.byref(a), _lib.stream_ptr()), 'ec_gemm')
    return out
