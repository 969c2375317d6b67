"""MFMA GEMM + fused epilogues against a plain torch fp32 reference (MI355X)."""
import pytest

pytestmark = pytest.mark.gpu


def ref_gemm(A, W, bias, epilogue, resid=None):
    import torch
    y = A.float() @ W.float().t()
    if bias is not None:
        y = y + bias
    if epilogue == 'gelu16':
        y = y * torch.sigmoid(1.702 * y)
    if epilogue == 'resid32':
        y = y + resid
    return y


@pytest.mark.parametrize('variant', [0])      # (the product library has the default kernel only)
@pytest.mark.parametrize('dt', ['float16', 'bfloat16'])
@pytest.mark.parametrize('M,N,K', [(257 * 3, 1024, 1024), (1000, 3072, 1024), (513, 1024, 4096),
                                   (77, 768, 640), (5, 512, 64), (256, 256, 128), (300, 48, 64)])
@pytest.mark.parametrize('epilogue', ['store16', 'gelu16', 'resid32', 'store32'])
def test_gemm_matches_torch(M, N, K, dt, epilogue, variant, hip):
    import torch
    from eventclip_amd import ops
    dtype = getattr(torch, dt)
    g = torch.Generator(device='cuda').manual_seed(M * 7 + N * 3 + K)
    A = torch.randn(M, K, device='cuda', generator=g).to(dtype)
    W = (torch.randn(N, K, device='cuda', generator=g) / K ** 0.5).to(dtype)
    bias = torch.randn(N, device='cuda', generator=g)
    resid = torch.randn(M, N, device='cuda', generator=g)
    out = resid.clone() if epilogue == 'resid32' else None
    got = ops.gemm(A, W, bias, epilogue, out=out, variant=variant)
    want = ref_gemm(A, W, bias, epilogue, resid)
    torch.cuda.synchronize()
    # inputs are identical 16-bit values on both sides; fp32 accumulation order differs,
    # 16-bit outputs add one rounding (2^-11 f16, 2^-8 bf16)
    out16 = epilogue in ('store16', 'gelu16')
    rtol = (2e-3 if dt == 'float16' else 1.6e-2) if out16 else 1e-4
    torch.testing.assert_close(got.float(), want, rtol=rtol, atol=rtol)


@pytest.mark.parametrize('variant', [0])
@pytest.mark.parametrize('epilogue', ['store16', 'gelu16', 'resid32', 'store32'])
@pytest.mark.parametrize('M,N,K', [(70001, 512, 256), (66000, 1024, 64), (40000, 784, 192)])
def test_gemm_many_tiles(M, N, K, epilogue, variant, hip):
    """More output tiles than CUs: several rounds of workgroups / a persistent workgroup's tile loop."""
    import torch
    from eventclip_amd import ops
    g = torch.Generator(device='cuda').manual_seed(M + N + K)
    A = torch.randn(M, K, device='cuda', generator=g).half()
    W = (torch.randn(N, K, device='cuda', generator=g) / K ** 0.5).half()
    bias = torch.randn(N, device='cuda', generator=g)
    resid = torch.randn(M, N, device='cuda', generator=g)
    out = resid.clone() if epilogue == 'resid32' else None
    got = ops.gemm(A, W, bias, epilogue, out=out, variant=variant)
    want = ref_gemm(A, W, bias, epilogue, resid)
    rtol = 2e-3 if epilogue in ('store16', 'gelu16') else 1e-4
    torch.testing.assert_close(got.float(), want, rtol=rtol, atol=rtol)


def test_gemm_exact_integer_layout(hip):
    """A = I-like selector with an asymmetric integer W catches swapped fragment maps."""
    import torch
    from eventclip_amd import ops
    M, N, K = 256, 256, 256
    A = torch.zeros(M, K, device='cuda', dtype=torch.float16)
    A[torch.arange(M), (torch.arange(M) * 7) % K] = 1.
    W = ((torch.arange(N, device='cuda')[:, None] * 3 + torch.arange(K, device='cuda')[None] * 5)
         % 61).to(torch.float16)
    got = ops.gemm(A, W, None, 'store32')
    want = A.float() @ W.float().t()
    assert torch.equal(got, want)


def test_gemm_no_bias_and_strided_A(hip):
    import torch
    from eventclip_amd import ops
    big = torch.randn(64, 2048, device='cuda').half()
    A = big[:, :1024]                      # lda = 2048
    W = torch.randn(768, 1024, device='cuda').half() / 32
    got = ops.gemm(A, W, None, 'store32')
    torch.testing.assert_close(got, A.float() @ W.float().t(), rtol=1e-4, atol=1e-4)


def test_gemm_rejects_bad_shapes(hip):
    import torch
    from eventclip_amd import ops
    A = torch.zeros(8, 100, device='cuda', dtype=torch.float16)
    W = torch.zeros(64, 100, device='cuda', dtype=torch.float16)
    with pytest.raises(RuntimeError):
        ops.gemm(A, W)


@pytest.mark.parametrize('M,N,K', [(257, 1024, 1024), (50, 768, 3072), (300, 3072, 1024), (1028, 4096, 1024), (7, 64, 128)])
@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
def test_low_latency_k_batched_launch_matches_the_single_pass(hip, M, N, K, dtype):
    """With scratch (ec_gemm_args.ws) an under-filled launch is cut into K-batches + a fixup kernel (serving latency);
    every epilogue gives the single-pass result up to the fp32 summation order."""
    import torch
    from eventclip_amd import ops
    dtype = getattr(torch, dtype)
    torch.manual_seed(M + N)
    A = torch.randn(M, K, device='cuda').to(dtype)
    W = (torch.randn(N, K, device='cuda') / K ** 0.5).to(dtype)
    bias = torch.randn(N, device='cuda')
    ws = torch.empty(80 << 20, dtype=torch.uint8, device='cuda')
    tol = dict(rtol=2e-3, atol=2e-3) if dtype == torch.float16 else dict(rtol=1.6e-2, atol=1.6e-2)
    for epi in ('store16', 'gelu16', 'store32'):
        torch.testing.assert_close(ops.gemm(A, W, bias, epi, ws=ws).float(), ops.gemm(A, W, bias, epi).float(), **tol)
    resid = torch.randn(M, N, device='cuda')
    torch.testing.assert_close(ops.gemm(A, W, bias, 'resid32', resid=resid, ws=ws), ops.gemm(A, W, bias, 'resid32', resid=resid),
                               rtol=1e-4, atol=1e-4)
    x1, x2 = resid.clone(), resid.clone()
    ops.gemm(A, W, bias, 'resid32', out=x1, ws=ws), ops.gemm(A, W, bias, 'resid32', out=x2)
    torch.testing.assert_close(x1, x2, rtol=1e-4, atol=1e-4)
    u1, u2 = torch.empty(M, N, device='cuda', dtype=dtype), torch.empty(M, N, device='cuda', dtype=dtype)
    g1, g2 = ops.gemm(A, W, bias, 'gelu16_save', aux=u1, ws=ws), ops.gemm(A, W, bias, 'gelu16_save', aux=u2)
    torch.testing.assert_close(g1.float(), g2.float(), **tol)
    torch.testing.assert_close(u1.float(), u2.float(), **tol)
    torch.testing.assert_close(ops.gemm(A, W, None, 'gelu_bwd16', aux=u2, ws=ws).float(),
                               ops.gemm(A, W, None, 'gelu_bwd16', aux=u2).float(), **tol)
    # a launch that fills the chip (tiles * 2 > 256 CUs for every N here) ignores the scratch: same bits
    big = torch.randn(256 * 96, K, device='cuda').to(dtype)
    assert torch.equal(ops.gemm(big, W, bias, 'store16', ws=ws), ops.gemm(big, W, bias, 'store16'))


def test_only_the_default_kernel_is_in_the_product_library(hip):
    """The comparison kernels (variants 1, 2, 3, 5) and the stamp / timeline variants live in the diagnostic
    build; the product's ec_gemm refuses them instead of running code nothing else exercises."""
    import torch
    from eventclip_amd import ops
    A = torch.randn(256, 64, device='cuda').half()
    W = torch.randn(256, 64, device='cuda').half()
    for v in (1, 2, 3, 5, 10, 16):
        with pytest.raises(RuntimeError, match='unknown variant'):
            ops.gemm(A, W, None, 'store16', variant=v)


# ------------------------------------------------------------------------------------------------
# LayerNorm folded into the GEMMs around it (EC_EPI_RESID_HL / ec_row_stats / EC_EPI_STORE16_LN / GELU16_LN)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('dt', ['float16', 'bfloat16'])
@pytest.mark.parametrize('M,N,K', [(257 * 3, 1024, 1024), (1000, 768, 3072), (77, 512, 64), (5, 256, 128), (300, 48, 64)])
def test_residual_update_on_hi_lo_planes(M, N, K, dt, hip):
    """(hi, lo) <- split(hi + lo + A W^T + b): the planes reproduce the fp32 residual update to ~2^-22 of the
    stream's magnitude, hi is the stream rounded to the operand type (what the next GEMM reads)."""
    import torch
    from eventclip_amd import ops
    dtype = getattr(torch, dt)
    g = torch.Generator(device='cuda').manual_seed(M + N + K)
    A = torch.randn(M, K, device='cuda', generator=g).to(dtype)
    W = (torch.randn(N, K, device='cuda', generator=g) / K ** 0.5).to(dtype)
    bias = torch.randn(N, device='cuda', generator=g)
    x = torch.randn(M, N, device='cuda', generator=g) * 4 + 0.5
    hi = x.to(dtype)
    lo = (x - hi.float()).half()
    want = hi.float() + lo.float() + A.float() @ W.float().t() + bias
    ops.gemm(A, W, bias, 'resid_hl', out=hi, aux=lo)
    got = hi.float() + lo.float()
    assert float((got - want).abs().max()) < 1e-6 * float(want.abs().max()) + 2e-5     # fp32 summation order + 2^-22
    assert float((hi.float() - want).abs().max()) <= (2 ** -10 if dt == 'float16' else 2 ** -7) * float(want.abs().max())


@pytest.mark.parametrize('dt', ['float16', 'bfloat16'])
@pytest.mark.parametrize('rows,width,stride', [(301, 1024, 1024), (9, 768, 5 * 768), (64, 64, 64), (1000, 1280, 1280)])
def test_row_stats(rows, width, stride, dt, hip):
    import torch
    from eventclip_amd import ops
    dtype = getattr(torch, dt)
    buf = (torch.randn(rows, stride, device='cuda') * 3 + 0.7).to(dtype)
    x = buf[:, :width]
    st = ops.row_stats(x)
    x32 = x.float()
    rstd = (x32.var(1, unbiased=False) + 1e-5).rsqrt()
    torch.testing.assert_close(st[:, 0], rstd, rtol=1e-5, atol=0)
    torch.testing.assert_close(st[:, 1], -rstd * x32.mean(1), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('dt', ['float16', 'bfloat16'])
@pytest.mark.parametrize('epilogue', ['store16_ln', 'gelu16_ln'])
@pytest.mark.parametrize('M,N,K', [(257 * 3, 3072, 1024), (1000, 4096, 1024), (77, 768, 640), (5, 512, 64), (300, 48, 128)])
def test_layernorm_finished_in_the_gemm_epilogue(M, N, K, epilogue, dt, hip):
    """LN(x) W^T + b from the RAW rows: A = x (16 bit), W' = W diag(gamma) (rounded once), and
    rstd (x W'^T) - rstd mean colsum(W') + (b + W beta) in the epilogue -- against fp32 LayerNorm + linear on the
    same x, and no further from it than LayerNorm -> 16 bit -> the plain GEMM."""
    import torch
    from eventclip_amd import ops
    dtype = getattr(torch, dt)
    g = torch.Generator(device='cuda').manual_seed(M * 3 + N + K)
    x = (torch.randn(M, K, device='cuda', generator=g) * 2 + 0.4).to(dtype)
    gamma = 1 + 0.2 * torch.randn(K, device='cuda', generator=g)
    beta = 0.3 * torch.randn(K, device='cuda', generator=g)
    Wt = torch.randn(N, K, device='cuda', generator=g) / K ** 0.5
    bias = 0.1 * torch.randn(N, device='cuda', generator=g)
    Wp = (Wt * gamma[None, :]).to(dtype)
    cs, bf = Wp.float().sum(1).contiguous(), (bias + Wt @ beta).contiguous()
    ref = torch.nn.functional.layer_norm(x.float(), (K,), gamma, beta, 1e-5) @ Wt.t() + bias
    act = (lambda t: t * torch.sigmoid(1.702 * t)) if epilogue == 'gelu16_ln' else (lambda t: t)
    got = ops.gemm(x, Wp, bf, epilogue, row_stats=ops.row_stats(x), col_sums=cs)
    h = torch.nn.functional.layer_norm(x.float(), (K,), gamma, beta, 1e-5).to(dtype)
    plain = ops.gemm(h, Wt.to(dtype), bias, 'gelu16' if epilogue == 'gelu16_ln' else 'store16')
    mag = float(act(ref).abs().max())
    e_fold, e_plain = float((got.float() - act(ref)).abs().max()) / mag, float((plain.float() - act(ref)).abs().max()) / mag
    assert e_fold < (2e-3 if dt == 'float16' else 1.6e-2) and e_fold < 1.5 * e_plain + 1e-4, (e_fold, e_plain)


def test_folded_epilogues_on_the_class_token_rows(hip):
    """The strided forms the last block uses: A rows S * W apart, statistics S rows apart, planes updated in place
    at row stride S * W -- the same values as the dense call on the gathered rows."""
    import torch
    from eventclip_amd import ops
    torch.manual_seed(3)
    n, S, W = 7, 5, 256
    x = (torch.randn(n * S, W, device='cuda') * 2).half()
    gamma, beta = 1 + 0.1 * torch.randn(W, device='cuda'), 0.1 * torch.randn(W, device='cuda')
    Wt = torch.randn(3 * W, W, device='cuda') / W ** 0.5
    Wp, bias = (Wt * gamma[None, :]).half(), 0.1 * torch.randn(3 * W, device='cuda')
    cs, bf = Wp.float().sum(1).contiguous(), (bias + Wt @ beta).contiguous()
    st_all = ops.row_stats(x)
    cls_rows = x.view(n, S * W)[:, :W]                       # row stride S * W
    got = ops.gemm(cls_rows, Wp, bf, 'store16_ln', row_stats=st_all, col_sums=cs, row_stats_stride=S)
    dense = x.view(n, S, W)[:, 0].contiguous()
    want = ops.gemm(dense, Wp, bf, 'store16_ln', row_stats=ops.row_stats(dense), col_sums=cs)
    assert torch.equal(got, want)
    # the residual update of the class rows only, in place in the planes
    A = torch.randn(n, 128, device='cuda').half()
    Wo, bo = (torch.randn(W, 128, device='cuda') * 0.1).half(), torch.randn(W, device='cuda')
    hi, lo = x.clone(), torch.zeros_like(x)
    ops.gemm(A, Wo, bo, 'resid_hl', out=hi.view(n, S * W)[:, :W], aux=lo.view(n, S * W)[:, :W])
    want = x.view(n, S, W)[:, 0].float() + A.float() @ Wo.float().t() + bo
    got = hi.view(n, S, W)[:, 0].float() + lo.view(n, S, W)[:, 0].float()
    assert float((got - want).abs().max()) < 1e-5 * float(want.abs().max())
    assert torch.equal(hi.view(n, S, W)[:, 1:], x.view(n, S, W)[:, 1:])      # the other tokens untouched


@pytest.mark.parametrize('dt', ['float16', 'bfloat16'])
@pytest.mark.parametrize('M,N,K', [(257 * 3, 1024, 1024), (1000, 768, 3072), (77, 512, 64), (5, 256, 128), (300, 1280, 64)])
def test_row_sums_out_of_the_residual_epilogue(M, N, K, dt, hip):
    """EC_EPI_RESID_HL with row_sums: per 64-column group (sum, sum of squares) of the NEW hi plane; merged, they are
    the LayerNorm statistics ec_row_stats reads off the plane."""
    import torch
    from eventclip_amd import ops
    dtype = getattr(torch, dt)
    g = torch.Generator(device='cuda').manual_seed(M + 2 * N + K)
    A = torch.randn(M, K, device='cuda', generator=g).to(dtype)
    W = (torch.randn(N, K, device='cuda', generator=g) / K ** 0.5).to(dtype)
    bias = torch.randn(N, device='cuda', generator=g)
    x = torch.randn(M, N, device='cuda', generator=g) * 3 + 0.8
    hi, lo = x.to(dtype), (x - x.to(dtype).float()).half()
    hi_ref, lo_ref = hi.clone(), lo.clone()
    ops.gemm(A, W, bias, 'resid_hl', out=hi_ref, aux=lo_ref)
    sums = torch.full((M, N // 64, 2), float('nan'), device='cuda')
    ops.gemm(A, W, bias, 'resid_hl', out=hi, aux=lo, row_sums=sums)
    assert torch.equal(hi, hi_ref) and torch.equal(lo, lo_ref)                 # the planes do not depend on the option
    h = hi.float().view(M, N // 64, 64)
    torch.testing.assert_close(sums[..., 0], h.sum(-1), rtol=1e-5, atol=1e-3)
    torch.testing.assert_close(sums[..., 1], (h * h).sum(-1), rtol=1e-5, atol=1e-3)
    st, want = ops.row_stats_merge(sums, N), ops.row_stats(hi)
    torch.testing.assert_close(st[:, 0], want[:, 0], rtol=2e-5, atol=0)
    torch.testing.assert_close(st[:, 1], want[:, 1], rtol=1e-4, atol=2e-5)


# ------------------------------------------------------------------------------------------------
# Split-precision operands in ONE launch (ec_gemm_args.A_lo / W_lo / row_sums_x, round 5)
# ------------------------------------------------------------------------------------------------
def _split(t, dtype):
    """(hi, lo) = (round16(t), round16(t - hi)), slices of one allocation as the packer keeps them."""
    import torch
    pair = torch.empty((2,) + tuple(t.shape), dtype=dtype, device=t.device)
    pair[0] = t.to(dtype)
    pair[1] = (t - pair[0].float()).to(dtype)
    return pair[0], pair[1]


@pytest.mark.parametrize('dt', ['float16', 'bfloat16'])
@pytest.mark.parametrize('parts', ['a_lo', 'w_lo', 'both'])
@pytest.mark.parametrize('epilogue', ['store32', 'resid32', 'store16', 'gelu16'])
@pytest.mark.parametrize('M,N,K', [(257 * 3, 1024, 1024), (513, 1024, 4096), (77, 768, 640), (5, 512, 64), (300, 48, 128)])
def test_split_operands_in_one_launch(M, N, K, epilogue, parts, dt, hip):
    """C = epi(A_lo W^T + A W_lo^T + A W^T + b) out of one launch: against the float64 product of the FULL operands
    (hi + lo) -- fp32 accuracy out of 16-bit MFMAs -- and bit-identical to the same call with a zero lo part in place
    of the missing one (a product with zeros adds nothing to the accumulators)."""
    import torch
    from eventclip_amd import ops
    dtype = getattr(torch, dt)
    g = torch.Generator(device='cuda').manual_seed(M + 3 * N + 5 * K)
    a = torch.randn(M, K, device='cuda', generator=g)
    w = torch.randn(N, K, device='cuda', generator=g) / K ** 0.5
    bias = torch.randn(N, device='cuda', generator=g)
    resid = torch.randn(M, N, device='cuda', generator=g)
    a_hi, a_lo = _split(a, dtype)
    w_hi, w_lo = _split(w, dtype)
    use_a, use_w = parts != 'w_lo', parts != 'a_lo'
    fa = a_hi.double() + (a_lo.double() if use_a else 0)
    fw = w_hi.double() + (w_lo.double() if use_w else 0)
    want = fa @ fw.t() + bias.double() - ((a_lo.double() @ w_lo.double().t()) if use_a and use_w else 0)   # (lo . lo is left out)
    if epilogue == 'resid32':
        want = want + resid.double()
    if epilogue == 'gelu16':
        want = want * torch.sigmoid(1.702 * want)
    out = resid.clone() if epilogue == 'resid32' else None
    got = ops.gemm(a_hi, w_hi, bias, epilogue, out=out, A_lo=a_lo if use_a else None, W_lo=w_lo if use_w else None)
    tol = 2e-6 if epilogue in ('store32', 'resid32') else (1e-3 if dt == 'float16' else 8e-3)
    err = float((got.double() - want).abs().max() / want.abs().max())
    assert err < tol, err
    # a zero lo part in place of the missing one: the same bits
    if parts != 'both':
        za, zw = torch.zeros_like(a_lo), torch.zeros_like(w_lo)
        pa = torch.stack([a_hi, a_lo if use_a else za])
        pw = torch.stack([w_hi, w_lo if use_w else zw])
        out2 = resid.clone() if epilogue == 'resid32' else None
        got2 = ops.gemm(pa[0], pw[0], bias, epilogue, out=out2, A_lo=pa[1], W_lo=pw[1])
        assert torch.equal(got, got2)


def test_split_operands_many_tiles_and_part_order(hip):
    """More tiles than CUs (the persistent loop's hand-over across a segment change), the lo part BELOW the hi part in
    memory, and parts further apart than a 32-bit offset reaches."""
    import torch
    from eventclip_amd import ops
    g = torch.Generator(device='cuda').manual_seed(9)
    M, N, K = 70001, 512, 256
    a = torch.randn(M, K, device='cuda', generator=g)
    w = torch.randn(N, K, device='cuda', generator=g) / K ** 0.5
    pa = torch.empty((2, M, K), dtype=torch.float16, device='cuda')
    pa[1] = a.half()                                  # hi part ABOVE the lo part
    pa[0] = (a - pa[1].float()).half()
    far = torch.empty((1 << 32) + 4096, dtype=torch.float16, device='cuda')          # the parts 8 GiB apart
    w_hi, w_lo = far[:N * K].view(N, K), far[-N * K:].view(N, K)
    w_hi.copy_(w.half())
    w_lo.copy_((w - w_hi.float()).half())
    want = (pa[1].double() + pa[0].double()) @ (w_hi.double() + w_lo.double()).t() - pa[0].double() @ w_lo.double().t()
    got = ops.gemm(pa[1], w_hi, None, 'store32', A_lo=pa[0], W_lo=w_lo)
    assert float((got.double() - want).abs().max() / want.abs().max()) < 2e-6


@pytest.mark.parametrize('epilogue', ['store16', 'gelu16'])
@pytest.mark.parametrize('M,N,K', [(257 * 3, 3072, 1024), (77, 768, 640), (300, 48, 128)])
def test_split_output_of_the_16_bit_epilogues(M, N, K, epilogue, hip):
    """STORE16 / GELU16 with args.aux (split-operand launches): C = hi = round16(v), aux = lo = round16(v - hi) of the
    epilogue's fp32 value v -- hi + lo reproduces epi(a w^T + b) of the full operands to ~1e-5, and hi is the plain call's
    output up to the double rounding of near-ties (the plain epilogue converts the fused product, this one the fp32
    value both parts are taken from)."""
    import torch
    from eventclip_amd import ops
    g = torch.Generator(device='cuda').manual_seed(M * 5 + N + K)
    a_hi, a_lo = _split(torch.randn(M, K, device='cuda', generator=g), torch.float16)
    w_hi, w_lo = _split(torch.randn(N, K, device='cuda', generator=g) / K ** 0.5, torch.float16)
    bias = 0.1 * torch.randn(N, device='cuda', generator=g)
    plain = ops.gemm(a_hi, w_hi, bias, epilogue, A_lo=a_lo, W_lo=w_lo)
    out_lo = torch.full((M, N), float('nan'), dtype=torch.float16, device='cuda')
    out_hi = ops.gemm(a_hi, w_hi, bias, epilogue, A_lo=a_lo, W_lo=w_lo, aux=out_lo)
    diff = out_hi != plain
    assert float(diff.float().mean()) < 2e-3                                                # a near-tie here and there
    assert float((out_hi.float() - plain.float()).abs().max()) <= 2 ** -9 * float(plain.float().abs().max())    # ... by one ulp
    ref = (a_hi.double() + a_lo.double()) @ (w_hi.double() + w_lo.double()).t() + bias.double()
    if epilogue == 'gelu16':
        ref = ref * torch.sigmoid(1.702 * ref)
    err = float(((out_hi.double() + out_lo.double()) - ref).abs().max() / ref.abs().max())
    assert err < 2e-5, err
    with pytest.raises(RuntimeError, match='A_lo / W_lo'):       # the lo output exists in the segmented launches only
        ops.gemm(a_hi, w_hi, bias, epilogue, aux=out_lo)


def test_split_operand_edges(hip):
    """M = 0 is a no-op with split operands too; what the segmented launches do not take (splits, ws, resid, transposed
    operands, a lo output outside STORE16 / GELU16) is refused, not ignored."""
    import torch
    from eventclip_amd import ops
    a_hi, a_lo = _split(torch.randn(64, 128, device='cuda'), torch.float16)
    w_hi, w_lo = _split(torch.randn(32, 128, device='cuda') / 11, torch.float16)
    out = ops.gemm(a_hi[:0], w_hi, None, 'store32', A_lo=a_lo[:0], W_lo=w_lo)
    assert out.shape == (0, 32)
    with pytest.raises(RuntimeError, match='A_lo / W_lo'):
        ops.gemm(a_hi, w_hi, None, 'resid32', out=torch.zeros(64, 32, device='cuda'), resid=torch.zeros(64, 32, device='cuda'), A_lo=a_lo)
    with pytest.raises(RuntimeError, match='A_lo / W_lo'):
        ops.gemm(a_hi, w_hi, None, 'store32', A_lo=a_lo, ws=torch.empty(1 << 20, dtype=torch.uint8, device='cuda'))
    with pytest.raises(RuntimeError, match='A_lo / W_lo'):
        ops.gemm(a_hi, w_hi, None, 'gelu16_save', aux=torch.empty(64, 32, dtype=torch.float16, device='cuda'), A_lo=a_lo)
    # the three-product form against float64 on a shape that is one partial tile
    want = (a_hi.double() + a_lo.double()) @ (w_hi.double() + w_lo.double()).t() - a_lo.double() @ w_lo.double().t()
    got = ops.gemm(a_hi, w_hi, None, 'store32', A_lo=a_lo, W_lo=w_lo)
    assert float((got.double() - want).abs().max() / want.abs().max()) < 2e-6


@pytest.mark.parametrize('parts', ['l8', 'w8', 'both', 'l8+w_lo'])
@pytest.mark.parametrize('epilogue', ['store32', 'store16', 'gelu16'])
@pytest.mark.parametrize('M,N,K', [(257 * 3, 1024, 1024), (513, 1024, 4096), (77, 768, 768), (5, 512, 128), (300, 48, 256)])
def test_lo_products_on_the_fp8_matrix_path(M, N, K, epilogue, parts, hip):
    """Round 6: the lo products as e4m3 operands on v_mfma_scale_f32_16x16x128_f8f6f4 (ec_gemm_args.A_lo8 + W8 in place of
    A_lo W^T, A8 + W_lo8 in place of A W_lo^T), in front of the 16-bit product(s) in the SAME launch and accumulators.
    The kernel must compute EXACTLY the product of the dequantised operands (e4m3 x e4m3 products are exact in fp32; only
    the summation order differs): against float64 of  A W^T + dq(A_lo8) dq(W8)^T [+ dq(A8) dq(W_lo8)^T | + A W_lo^T] + b."""
    import torch
    from eventclip_amd import ops
    g = torch.Generator(device='cuda').manual_seed(M + 3 * N + 5 * K)
    a = torch.randn(M, K, device='cuda', generator=g)
    w = torch.randn(N, K, device='cuda', generator=g) / K ** 0.5
    bias = torch.randn(N, device='cuda', generator=g)
    a_hi, a_lo = _split(a, torch.float16)
    w_hi, w_lo = _split(w, torch.float16)
    kw = {}
    want = a_hi.double() @ w_hi.double().t() + bias.double()
    if parts in ('l8', 'both', 'l8+w_lo'):
        kw['A_lo8'] = ops.quantize_e4m3(a - a_hi.float(), exp=12)
        kw['W8'] = ops.quantize_e4m3(w_hi)
        want = want + ops.dequantize_e4m3(*kw['A_lo8'], K).double() @ ops.dequantize_e4m3(*kw['W8'], K).double().t()
    if parts in ('w8', 'both'):
        kw['A8'] = ops.quantize_e4m3(a_hi, exp=0)
        kw['W_lo8'] = ops.quantize_e4m3(w - w_hi.float())
        want = want + ops.dequantize_e4m3(*kw['A8'], K).double() @ ops.dequantize_e4m3(*kw['W_lo8'], K).double().t()
    if parts == 'l8+w_lo':
        kw['W_lo'] = w_lo
        want = want + a_hi.double() @ w_lo.double().t()
    if epilogue == 'gelu16':
        want = want * torch.sigmoid(1.702 * want)
    got = ops.gemm(a_hi, w_hi, bias, epilogue, **kw)
    tol = 2e-6 if epilogue == 'store32' else 1e-3
    err = float((got.double() - want).abs().max() / want.abs().max())
    assert err < tol, err
    if epilogue == 'store32':
        # ... and with them the result is close to the product of the FULL operands: the e4m3 lo products remove most of the
        # 16-bit operand rounding (3e-4 of the result without them; 2e-6 with 16-bit lo parts)
        full = a.double() @ w.double().t() + bias.double()
        plain = ops.gemm(a_hi, w_hi, bias, epilogue)
        e8 = float((got.double() - full).abs().max() / full.abs().max())
        e16 = float((plain.double() - full).abs().max() / full.abs().max())
        if parts == 'both' and K >= 256:
            assert e8 < 0.15 * e16, (e8, e16)


def test_fp8_lo_products_edges(hip):
    """What the e4m3 segments do not take is refused: bf16, K not a multiple of 128, a lo product given twice, a missing
    partner operand; the lo output (aux) and RESID_HL work with them; many tiles (hand-over across the e4m3 -> 16-bit
    segment change in the persistent loop)."""
    import torch
    from eventclip_amd import ops
    g = torch.Generator(device='cuda').manual_seed(3)
    M, N, K = 70001, 512, 256
    a = torch.randn(M, K, device='cuda', generator=g)
    w = torch.randn(N, K, device='cuda', generator=g) / K ** 0.5
    a_hi, a_lo = _split(a, torch.float16)
    w_hi, w_lo = _split(w, torch.float16)
    A_lo8, W8 = ops.quantize_e4m3(a - a_hi.float(), exp=12), ops.quantize_e4m3(w_hi)
    want = a_hi.double() @ w_hi.double().t() + ops.dequantize_e4m3(*A_lo8, K).double() @ ops.dequantize_e4m3(*W8, K).double().t()
    got = ops.gemm(a_hi, w_hi, None, 'store32', A_lo8=A_lo8, W8=W8)
    assert float((got.double() - want).abs().max() / want.abs().max()) < 2e-6
    # RESID_HL: the planes updated by the two-product sum
    hi = torch.randn(M, N, device='cuda', generator=g).half()
    lo = (torch.randn(M, N, device='cuda', generator=g) * 1e-4).half()
    ref = hi.double() + lo.double() + want
    ops.gemm(a_hi, w_hi, None, 'resid_hl', out=hi, aux=lo, A_lo8=A_lo8, W8=W8)
    assert float(((hi.double() + lo.double()) - ref).abs().max() / ref.abs().max()) < 2e-6
    # the lo output of STORE16
    out_lo = torch.empty(M, N, dtype=torch.float16, device='cuda')
    out_hi = ops.gemm(a_hi, w_hi, None, 'store16', aux=out_lo, A_lo8=A_lo8, W8=W8)
    assert float(((out_hi.double() + out_lo.double()) - want).abs().max() / want.abs().max()) < 2e-5
    with pytest.raises(RuntimeError, match='not both'):
        ops.gemm(a_hi, w_hi, None, 'store32', A_lo=a_lo, A_lo8=A_lo8, W8=W8)
    with pytest.raises(RuntimeError, match='needs W8'):
        ops.gemm(a_hi, w_hi, None, 'store32', A_lo8=A_lo8)
    with pytest.raises(RuntimeError, match='K %% 128|K % 128'):
        ops.gemm(a_hi[:, :192], w_hi[:, :192], None, 'store32', A_lo8=(A_lo8[0][:, :384], 12), W8=(W8[0][:, :384], W8[1]))
    b_hi = a_hi.bfloat16()
    with pytest.raises(RuntimeError, match='EC_F16'):
        ops.gemm(b_hi, w_hi.bfloat16(), None, 'store32', A_lo8=A_lo8, W8=W8)


@pytest.mark.parametrize('rows,width', [(771, 1024), (5, 768), (1030, 1280)])
def test_layernorm_hl8_writes_e4m3_operands(rows, width, hip):
    """ec_layernorm_hl8: LayerNorm of the hi + lo planes into the fp16 hi part (bit-identical to ec_layernorm_hl's), the lo
    part as e4m3 of lo . 2^12 and an e4m3 copy of the hi part, one byte per element at the 16-bit row pitch -- exactly
    torch's float8_e4m3fn rounding of the same fp32 values."""
    import ctypes
    import torch
    from eventclip_amd import _lib, ops
    g = torch.Generator(device='cuda').manual_seed(rows + width)
    x = torch.randn(rows, width, device='cuda', generator=g) * 3 + 0.5
    x[:, 7] *= 40                                                    # an outlier channel
    x_hi = x.half()
    x_lo = (x - x_hi.float()).half()
    gamma = 1 + 0.1 * torch.randn(width, device='cuda', generator=g)
    beta = 0.1 * torch.randn(width, device='cuda', generator=g)
    o_hi = torch.empty(rows, width, dtype=torch.float16, device='cuda')
    o_lo = torch.empty(rows, width, dtype=torch.float16, device='cuda')
    lib = _lib.lib()
    _lib.check(lib.ec_layernorm_hl(_lib.ptr(x_hi), _lib.ptr(x_lo), width, _lib.ptr(gamma), _lib.ptr(beta), rows, width, 1e-5,
                                   _lib.ptr(o_hi), _lib.ptr(o_lo), width, _lib.EC_F16, _lib.stream_ptr()))
    p_hi = torch.empty_like(o_hi)
    lo8 = torch.full((rows, 2 * width), 0xAA, dtype=torch.uint8, device='cuda')
    hi8 = torch.full((rows, 2 * width), 0xAA, dtype=torch.uint8, device='cuda')
    _lib.check(lib.ec_layernorm_hl8(_lib.ptr(x_hi), _lib.ptr(x_lo), width, _lib.ptr(gamma), _lib.ptr(beta), rows, width, 1e-5,
                                    _lib.ptr(p_hi), _lib.ptr(lo8), _lib.ptr(hi8), width, 12, 0, _lib.stream_ptr()))
    assert torch.equal(p_hi, o_hi)
    assert bool((lo8[:, width:] == 0xAA).all()) and bool((hi8[:, width:] == 0xAA).all())      # the second half of a row is not touched
    # the lo bytes: e4m3 of the SAME fp32 lo value the 16-bit kernel rounded to fp16 (o_lo is that value to 2^-11)
    lo_ref = ops.dequantize_e4m3(lo8, 12, width)
    err = (lo_ref - o_lo.float()).abs()
    assert bool((err <= 2.0 ** -4 * o_lo.float().abs() + 2.0 ** -21).all())
    want_hi8 = o_hi.float().clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    assert torch.equal(hi8[:, :width].contiguous(), want_hi8)
    # without the hi copy
    lo8b = torch.zeros_like(lo8)
    _lib.check(lib.ec_layernorm_hl8(_lib.ptr(x_hi), _lib.ptr(x_lo), width, _lib.ptr(gamma), _lib.ptr(beta), rows, width, 1e-5,
                                    _lib.ptr(p_hi), _lib.ptr(lo8b), None, width, 12, 0, _lib.stream_ptr()))
    assert torch.equal(lo8b[:, :width], lo8[:, :width])


@pytest.mark.parametrize('M,N,K', [(257 * 3, 4096, 1024), (77, 768, 768), (300, 48, 256)])
def test_gelu16_writes_its_lo_part_as_e4m3(M, N, K, hip):
    """EC_EPI_GELU16 with ec_gemm_args.aux_e4m3 (the c_fc GEMM of a split-operand block under lo_fp8): C = hi = round16(v),
    aux = round_e4m3((v - hi) . 2^12), one byte per element in the first N bytes of rows of 2 N bytes -- the A_lo8 operand of
    c_proj.  hi is bit-identical to the same launch with a 16-bit lo output, the e4m3 bytes are the rounding of that launch's
    lo part (to the 2^-11 the fp16 lo itself carries), the second half of every aux row is not touched, and hi + dq(lo8)
    reproduces QuickGELU(a w^T + b) of the dequantised operands to 2^-4 of a 16-bit ulp."""
    import torch
    from eventclip_amd import ops
    g = torch.Generator(device='cuda').manual_seed(M + N + K)
    a = torch.randn(M, K, device='cuda', generator=g)
    w = torch.randn(N, K, device='cuda', generator=g) / K ** 0.5
    bias = 0.1 * torch.randn(N, device='cuda', generator=g)
    a_hi, w_hi = a.half(), w.half()
    A_lo8, W8 = ops.quantize_e4m3(a - a_hi.float(), exp=12), ops.quantize_e4m3(w_hi)
    lo16 = torch.full((M, N), float('nan'), dtype=torch.float16, device='cuda')
    hi_a = ops.gemm(a_hi, w_hi, bias, 'gelu16', aux=lo16, A_lo8=A_lo8, W8=W8)
    lo8 = torch.full((M, 2 * N), 0x55, dtype=torch.uint8, device='cuda')
    hi_b = ops.gemm(a_hi, w_hi, bias, 'gelu16', aux8=(lo8, 12), A_lo8=A_lo8, W8=W8)
    assert torch.equal(hi_a, hi_b)
    assert bool((lo8[:, N:] == 0x55).all())
    dq = ops.dequantize_e4m3(lo8, 12, N)
    assert bool(((dq - lo16.float()).abs() <= 2.0 ** -4 * lo16.float().abs() + 2.0 ** -20).all())
    ref = a_hi.double() @ w_hi.double().t() + bias.double() + ops.dequantize_e4m3(*A_lo8, K).double() @ ops.dequantize_e4m3(*W8, K).double().t()
    ref = ref * torch.sigmoid(1.702 * ref)
    e8 = float(((hi_b.double() + dq.double()) - ref).abs().max() / ref.abs().max())
    e16 = float((hi_b.double() - ref).abs().max() / ref.abs().max())
    assert e8 < 0.12 * e16 + 2e-6, (e8, e16)
    with pytest.raises(RuntimeError, match='aux_e4m3'):        # the e4m3 lo output exists with e4m3 lo products and GELU16 only
        ops.gemm(a_hi, w_hi, bias, 'store16', aux8=(lo8, 12), A_lo8=A_lo8, W8=W8)
