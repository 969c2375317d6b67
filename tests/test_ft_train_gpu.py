"""Fine-tuning the vision tower on the MI355X: the building blocks against torch
fp32, the tower's gradients against the oracle's autograd, and whole optimisation steps against the
reference's own FTCLIPClassifier + torch.optim.Adam (tests/golden/ft_train.npz).

Tolerances.  The path is mixed precision as torch.cuda.amp is for the reference: 16-bit MFMA operands
(activations, weights, activation gradients), fp32 accumulation and residual-stream gradients.  A gradient
tensor is compared as a whole: relative l2 error <= GRAD_REL (2e-2; measured 2e-3 .. 8e-3) and cosine >= 0.9995
against fp32 / fp64 autograd."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

GRAD_REL = 2e-2


def rel_l2(got, want):
    got, want = got.double().flatten().cpu(), want.double().flatten().cpu()
    return float((got - want).norm() / want.norm().clamp_min(1e-30))


def cosine(got, want):
    got, want = got.double().flatten().cpu(), want.double().flatten().cpu()
    return float((got @ want) / (got.norm() * want.norm()).clamp_min(1e-30))


# ------------------------------------------------------------------------------------------------
# GEMM: the training epilogues and K-batches
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_gemm_residual_source_and_gelu_pair(hip, dtype):
    from eventclip_amd import ops
    torch.manual_seed(0)
    M, N, K = 777, 512, 256
    A = torch.randn(M, K, device='cuda').to(dtype)
    W = (torch.randn(N, K, device='cuda') / K ** 0.5).to(dtype)
    bias = torch.randn(N, device='cuda')
    ref = A.float() @ W.float().t() + bias
    # out = resid + acc, resid untouched
    resid = torch.randn(M, N, device='cuda')
    keep = resid.clone()
    out = ops.gemm(A, W, bias, 'resid32', resid=resid)
    assert torch.equal(resid, keep)
    torch.testing.assert_close(out, ref + resid, rtol=2e-3, atol=2e-3)
    # in-place form gives the same bits
    inplace = resid.clone()
    ops.gemm(A, W, bias, 'resid32', out=inplace)
    assert torch.equal(inplace, out)
    # activation + pre-activation: the activation is bit-identical to the inference epilogue
    u = torch.empty(M, N, device='cuda', dtype=dtype)
    g = ops.gemm(A, W, bias, 'gelu16_save', aux=u)
    assert torch.equal(g, ops.gemm(A, W, bias, 'gelu16'))
    assert torch.equal(u, ops.gemm(A, W, bias, 'store16'))
    # gradient through the activation: acc * QuickGELU'(u)
    d = ops.gemm(A, W, None, 'gelu_bwd16', aux=u)
    uf = u.float()
    sg = torch.sigmoid(1.702 * uf)
    want = (A.float() @ W.float().t()).to(dtype).float() * (sg * (1 + 1.702 * uf * (1 - sg)))
    tol = 2e-3 if dtype == torch.float16 else 1.6e-2
    torch.testing.assert_close(d.float(), want, rtol=tol, atol=tol)


@pytest.mark.parametrize('splits', [2, 4, 16])
def test_gemm_k_batches_are_the_partial_products(hip, splits):
    from eventclip_amd import ops
    torch.manual_seed(1)
    M, N, Kc = 300, 272, 128
    A = torch.randn(M, Kc * splits, device='cuda').half()
    W = torch.randn(N, Kc * splits, device='cuda').half()
    part = ops.gemm(A, W, None, 'store32', splits=splits)
    assert tuple(part.shape) == (splits, M, N)
    for s in range(splits):
        ref = A[:, s * Kc:(s + 1) * Kc].float() @ W[:, s * Kc:(s + 1) * Kc].float().t()
        torch.testing.assert_close(part[s], ref, rtol=1e-3, atol=1e-2)
    torch.testing.assert_close(part.sum(0), A.float() @ W.float().t(), rtol=1e-3, atol=3e-2)
    # a weight with a row stride (ldw): a column window of a wider matrix
    wide = torch.randn(N, Kc * splits + 64, device='cuda').half()
    win = wide[:, 64:]
    torch.testing.assert_close(ops.gemm(A, win, None, 'store32'), A.float() @ win.float().t(), rtol=1e-3, atol=3e-2)


# ------------------------------------------------------------------------------------------------
# attention
# ------------------------------------------------------------------------------------------------
def _attention(qkv, n, S, W, heads, lse=True):
    import ctypes
    from eventclip_amd import _lib, ops
    out = torch.empty(n * S, W, device='cuda', dtype=qkv.dtype)
    l2 = torch.empty(n, heads, S, device='cuda')
    rc = _lib.lib().ec_attention_train(_lib.ptr(qkv), _lib.ptr(out), _lib.ptr(l2), n, S, W, heads,
                                       ops.dtype_code(qkv.dtype), _lib.stream_ptr())
    _lib.check(rc, 'ec_attention_train')
    return out, l2


def _attention_bwd(qkv, out, l2, dout, n, S, W, heads):
    from eventclip_amd import _lib, ops
    dqkv = torch.full_like(qkv, float('nan'))
    delta = torch.empty(n, heads, S, device='cuda')
    rc = _lib.lib().ec_attention_backward(_lib.ptr(qkv), _lib.ptr(out), _lib.ptr(l2), _lib.ptr(dout), _lib.ptr(dqkv),
                                          _lib.ptr(delta), n, S, W, heads, ops.dtype_code(qkv.dtype), _lib.stream_ptr())
    _lib.check(rc, 'ec_attention_backward')
    return dqkv


@pytest.mark.parametrize('rows,M,N,splits,dtype', [
    (64, 256, 256, 1, torch.float16), (1000, 256, 512, 1, torch.float16), (77, 72, 80, 1, torch.bfloat16),
    (16448, 1024, 3072, 4, torch.float16), (16448, 4096, 1024, 2, torch.bfloat16), (5000, 768, 768, 8, torch.float16),
    (300, 8, 16, 2, torch.float16), (4100, 1024, 1024, 16, torch.float16)])
def test_gemm_over_the_rows_of_transposed_operands(hip, rows, M, N, splits, dtype):
    """ec_gemm_args.transposed: C = A^T W over the ROW index of two row-major operands (column slices of wider
    matrices here, as dq | dk | dv are), in row batches whose last K tile runs past the rows that exist; against
    fp32 matmul of the same 16-bit values, and bit-identical when repeated."""
    from eventclip_amd import ops
    torch.manual_seed(rows + M)
    wideA = torch.randn(rows, M + 64, device='cuda').to(dtype)
    wideW = torch.randn(rows, N + 32, device='cuda').to(dtype)
    A, W = wideA[:, 32:32 + M], wideW[:, 16:16 + N]
    got = ops.gemm_rows(A, W, splits)
    K = ((rows + splits - 1) // splits + 63) // 64 * 64
    want = A.float().T @ W.float()
    total = got if splits == 1 else got.sum(0)
    scale = float(want.abs().max())
    assert float((total - want).abs().max()) < 2e-5 * scale * max(1.0, rows / 1000) + 1e-3
    if splits > 1:
        for sidx in range(splits):
            lo, hi = sidx * K, min((sidx + 1) * K, rows)
            part = A[lo:hi].float().T @ W[lo:hi].float() if lo < rows else torch.zeros_like(want)
            assert float((got[sidx] - part).abs().max()) < 2e-5 * scale + 1e-3, sidx
    assert torch.equal(got, ops.gemm_rows(A, W, splits))


@pytest.mark.parametrize('S,heads,n', [(5, 1, 3), (50, 2, 4), (197, 2, 2), (257, 3, 3), (577, 2, 2)])
@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_attention_forward_saves_lse_and_backward_matches_autograd(hip, S, heads, n, dtype):
    from eventclip_amd import _lib, ops
    torch.manual_seed(S)
    W = heads * 64
    qkv = (torch.randn(n * S, 3 * W, device='cuda') * 1.5).to(dtype)
    out, l2 = _attention(qkv, n, S, W, heads)
    # same output bits as the inference entry point
    plain = torch.empty_like(out)
    _lib.check(_lib.lib().ec_attention(_lib.ptr(qkv), _lib.ptr(plain), n, S, W, heads, 0, ops.dtype_code(dtype),
                                       _lib.stream_ptr()), 'ec_attention')
    assert torch.equal(out, plain)
    x = qkv.float().requires_grad_(True)
    q, k, v = [t.view(n, S, heads, 64).transpose(1, 2) for t in x.view(n, S, 3 * W).split(W, dim=-1)]
    sc = (q @ k.transpose(-1, -2)) * 0.125
    ref = (sc.softmax(-1) @ v).transpose(1, 2).reshape(n * S, W)
    # the row sum is that of the ROUNDED probabilities the output was built from (matrix pipe); with the running
    # maximum left where it is, a row's largest P is 2^x rounded to the operand type instead of exactly 1:
    # up to 2^-9 relative = 2.8e-3 in log2 units for bf16 (f16: 3.5e-4)
    torch.testing.assert_close(l2, torch.logsumexp(sc.detach(), -1) * 1.4426950408889634, rtol=1e-4,
                               atol=2e-3 if dtype == torch.float16 else 4e-3)
    dout = (torch.randn(n * S, W, device='cuda') * 0.5).to(dtype)
    ref.backward(dout.float())
    got = _attention_bwd(qkv, out, l2, dout, n, S, W, heads)
    assert torch.isfinite(got.float()).all()
    tol = 6e-3 if dtype == torch.float16 else 4e-2
    for j, name in enumerate(('dq', 'dk', 'dv')):
        a, b = got[:, j * W:(j + 1) * W].float(), x.grad[:, j * W:(j + 1) * W]
        assert rel_l2(a, b) < tol, (name, rel_l2(a, b))


# ------------------------------------------------------------------------------------------------
# LayerNorm backward, weight packing, small products
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('rows,width,stride', [(1, 64, 64), (37, 768, 768), (1028, 1024, 1024), (9, 1024, 5 * 1024)])
def test_layernorm_backward_matches_autograd(hip, rows, width, stride):
    from eventclip_amd import _lib
    torch.manual_seed(rows)
    xbuf = torch.randn(rows, stride, device='cuda') * 2 + 0.3
    dy = torch.randn(rows, width, device='cuda')
    gamma = 1 + 0.2 * torch.randn(width, device='cuda')
    beta = torch.randn(width, device='cuda')
    x = xbuf[:, :width].clone().requires_grad_(True)
    gp = gamma.clone().requires_grad_(True)
    bp = beta.clone().requires_grad_(True)
    F.layer_norm(x, (width,), gp, bp, 1e-5).backward(dy)
    base = torch.randn(rows, width, device='cuda')
    for accumulate in (0, 1):
        dx = base.clone()
        dg, db = torch.empty(width, device='cuda'), torch.empty(width, device='cuda')
        part = torch.empty(int(_lib.lib().ec_layernorm_backward_partials(rows, width)), device='cuda')
        rc = _lib.lib().ec_layernorm_backward(_lib.ptr(xbuf), stride, _lib.ptr(dy), width, _lib.ptr(gamma), rows, width,
                                              1e-5, _lib.ptr(dx), width, accumulate, _lib.ptr(dg), _lib.ptr(db),
                                              _lib.ptr(part), _lib.stream_ptr())
        _lib.check(rc, 'ec_layernorm_backward')
        torch.testing.assert_close(dx, x.grad + (base if accumulate else 0), rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(dg, gp.grad, rtol=1e-4, atol=1e-3)
        torch.testing.assert_close(db, bp.grad, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_pack_weight16(hip, dtype):
    from eventclip_amd import ft, ops
    torch.manual_seed(2)
    w = torch.randn(200, 136, device='cuda')
    hi, lo = torch.empty(200, 136, device='cuda', dtype=dtype), torch.empty(200, 136, device='cuda', dtype=dtype)
    hi_t = torch.empty(136, 200, device='cuda', dtype=dtype)
    ft.pack_weight16(w, ops.dtype_code(dtype), hi=hi, lo=lo, hi_t=hi_t)
    assert torch.equal(hi, w.to(dtype))
    assert torch.equal(lo, (w - w.to(dtype).float()).to(dtype))
    assert torch.equal(hi_t, w.to(dtype).t().contiguous())
    # a list of same-shape matrices in one launch, outputs optional per matrix
    ws = [torch.randn(200, 136, device='cuda') for _ in range(3)]
    his = [torch.empty(200, 136, device='cuda', dtype=dtype) for _ in ws]
    hts = [torch.empty(136, 200, device='cuda', dtype=dtype) for _ in ws]
    ft.pack_weights16([(ws[0], his[0], None, hts[0]), (ws[1], None, None, hts[1]), (ws[2], his[2], None, None)],
                      ops.dtype_code(dtype))
    torch.cuda.synchronize()
    assert torch.equal(his[0], ws[0].to(dtype)) and torch.equal(hts[1], ws[1].to(dtype).t().contiguous())
    assert torch.equal(his[2], ws[2].to(dtype)) and torch.equal(hts[0], ws[0].to(dtype).t().contiguous())


def test_sgemm_any_layout(hip):
    from eventclip_amd import ft
    torch.manual_seed(3)
    a, b = torch.randn(70, 33, device='cuda'), torch.randn(33, 130, device='cuda')
    out = torch.randn(70, 130, device='cuda')
    want = 0.5 * a @ b + 2.0 * out
    torch.testing.assert_close(ft.sgemm(a, b, out, 0.5, 2.0), want, rtol=1e-5, atol=1e-5)
    at, bt = torch.randn(33, 70, device='cuda'), torch.randn(130, 33, device='cuda')
    o2 = torch.empty(70, 130, device='cuda')
    torch.testing.assert_close(ft.sgemm(at.t(), bt.t(), o2), at.t() @ bt.t(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('rows,cols,r,n_items', [(64, 64, 2, 1), (1024, 1024, 16, 5), (768, 768, 4, 3), (200, 136, 64, 2)])
def test_lora_merge_and_chain_rule(hip, rows, cols, r, n_items):
    from eventclip_amd import _lib, ft
    torch.manual_seed(r)
    mk = lambda *shape: [torch.randn(*shape, device='cuda') for _ in range(n_items)]      # noqa: E731
    base, dW, up, down = mk(rows, cols), mk(rows, cols), mk(rows, r), mk(r, cols)
    out, d_up, d_down = mk(rows, cols), mk(rows, r), mk(r, cols)
    items = (_lib.EcLoraItem * n_items)()
    for i, it in enumerate(items):
        it.base, it.up, it.down, it.out = base[i].data_ptr(), up[i].data_ptr(), down[i].data_ptr(), out[i].data_ptr()
        it.dW, it.d_up, it.d_down = dW[i].data_ptr(), d_up[i].data_ptr(), d_down[i].data_ptr()
    table = ft.device_table(items)
    _lib.check(_lib.lib().ec_lora_merge_batched(_lib.ptr(table), n_items, rows, cols, r, _lib.stream_ptr()), 'merge')
    scratch = torch.empty(int(_lib.lib().ec_lora_grad_scratch_floats(n_items, rows, cols, r)), device='cuda')
    _lib.check(_lib.lib().ec_lora_grad_batched(_lib.ptr(table), n_items, rows, cols, r, _lib.ptr(scratch),
                                               _lib.stream_ptr()), 'grad')
    for i in range(n_items):
        torch.testing.assert_close(out[i], base[i] + up[i] @ down[i], rtol=1e-5, atol=1e-4)
        torch.testing.assert_close(d_up[i], dW[i] @ down[i].t(), rtol=1e-4, atol=1e-3)
        torch.testing.assert_close(d_down[i], up[i].t() @ dW[i], rtol=1e-4, atol=1e-3)


def test_adam_over_a_tensor_list_matches_torch(hip):
    from eventclip_amd import _lib, ft
    torch.manual_seed(9)
    shapes = [(1000, 37), (5,), (64, 64), (3, 1, 7)]
    params = [torch.randn(*s, device='cuda') for s in shapes]
    ref = [p.clone().requires_grad_(True) for p in params]
    opt = torch.optim.Adam([{'params': ref[:2], 'lr': 1e-2}, {'params': ref[2:], 'lr': 3e-3}])
    grads = [torch.zeros_like(p) for p in params]
    m, v = [torch.zeros_like(p) for p in params], [torch.zeros_like(p) for p in params]
    items = (_lib.EcAdamItem * len(params))()
    for i, it in enumerate(items):
        it.param, it.grad, it.exp_avg, it.exp_avg_sq = params[i].data_ptr(), grads[i].data_ptr(), m[i].data_ptr(), v[i].data_ptr()
        it.n, it.group = params[i].numel(), int(i >= 2)
    table = ft.device_table(items)
    for step in (1, 2, 3):
        for g, r in zip(grads, ref):
            g.copy_(torch.randn_like(g))
            r.grad = g.clone()
        opt.step()
        rc = _lib.lib().ec_adam_step_multi(_lib.ptr(table), len(params), max(p.numel() for p in params), 1e-2, 3e-3, 0.9,
                                           0.999, 1e-8, 0., step, None, None, _lib.stream_ptr())
        _lib.check(rc, 'ec_adam_step_multi')
    for p, r in zip(params, ref):
        torch.testing.assert_close(p, r.detach(), rtol=1e-5, atol=1e-6)
    # a raised skip flag leaves everything untouched
    flag = torch.ones(1, dtype=torch.int32, device='cuda')
    keep = [p.clone() for p in params]
    _lib.check(_lib.lib().ec_adam_step_multi(_lib.ptr(table), len(params), max(p.numel() for p in params), 1e-2, 3e-3, 0.9,
                                             0.999, 1e-8, 0., 4, _lib.ptr(flag), None, _lib.stream_ptr()), 'ec_adam_step_multi')
    assert all(torch.equal(a, b) for a, b in zip(keep, params))


# ------------------------------------------------------------------------------------------------
# the tower: forward with a tape, gradients of every parameter
# ------------------------------------------------------------------------------------------------
def _tiny_cfg(**kw):
    cfg = dict(image_size=8, patch=4, width=64, layers=2, embed_dim=16, text_width=64, text_heads=1, text_layers=1,
               context_length=77, vocab_size=128)
    cfg.update(kw)
    return cfg


CONFIGS = {
    'tiny': (_tiny_cfg(), 5),
    'b32_2blocks': (_tiny_cfg(image_size=224, patch=32, width=768, layers=2, embed_dim=512), 6),
    'l14_2blocks': (_tiny_cfg(image_size=224, patch=14, width=1024, layers=2, embed_dim=768), 3),
    'wide_odd': (_tiny_cfg(image_size=48, patch=16, width=128, layers=3, embed_dim=32), 9),
    # S = 577: the attention backward stages keys / queries in chunks of 288 rows
    'l14_336_1block': (_tiny_cfg(image_size=336, patch=14, width=1024, layers=1, embed_dim=768), 2),
}


def _tower(cfg, seed=0, dtype='float16'):
    from eventclip_amd import clip as eclip, ft
    sd = eclip.random_state_dict(cfg, seed=seed)
    # the inference tower packed the way the training path holds its weights (plain q, LayerNorm as its own launch):
    # the two forwards are then the same kernels on the same operand copies
    model = eclip.CLIP(cfg, sd, dtype=dtype, full_last_block=True, ln_folded=False, q_scaled=False).cuda()
    return model, ft.VisualTower(model), sd


def _patchify(tower, imgs):
    from eventclip_amd import _lib
    x = imgs.cuda().float().contiguous()
    patches = torch.empty((x.shape[0], tower.G, tower.kpad), dtype=tower.cd, device='cuda')
    _lib.check(_lib.lib().ec_patchify(_lib.ptr(x), x.shape[0], tower.cfg['image_size'], tower.P, tower.kpad,
                                      _lib.ptr(patches), tower.code, _lib.stream_ptr()), 'ec_patchify')
    return patches


@pytest.mark.parametrize('name', list(CONFIGS))
def test_training_forward_is_the_inference_forward(hip, name):
    cfg, n = CONFIGS[name]
    model, tower, _ = _tower(cfg)
    torch.manual_seed(4)
    imgs = torch.randn(n, 3, cfg['image_size'], cfg['image_size'])
    patches = _patchify(tower, imgs)
    feats = tower.forward(patches)
    want = model.encode_patches(patches)
    assert torch.equal(feats, want)             # same kernels, same operand copies: same bits
    assert torch.equal(tower.encode_patches(patches), want)


@pytest.mark.parametrize('name', list(CONFIGS))
@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
def test_tower_gradients_match_the_oracle(hip, name, dtype):
    from oracle import clip_ref
    if dtype == 'bfloat16' and name not in ('tiny', 'wide_odd'):
        pytest.skip('bf16 operands: covered on the small towers')
    cfg, n = CONFIGS[name]
    model, tower, sd = _tower(cfg, seed=1, dtype=dtype)
    torch.manual_seed(5)
    R = cfg['image_size']
    imgs = torch.randn(n, 3, R, R)
    d_feats = torch.randn(n, cfg['embed_dim'])
    # oracle: fp64 autograd through the functional tower (the effective 16-bit weights are a separate question:
    # the comparison is against the fp32 masters, as the reference trains)
    leaves = {k: v.double().clone().requires_grad_(True) for k, v in sd.items() if k.startswith('visual.')}
    clip_ref.encode_image_autograd(leaves, cfg, imgs.double()).backward(d_feats.double())
    want = list(tower.master)
    tower.forward(_patchify(tower, imgs))
    grads, flat = tower.backward(d_feats.cuda(), want)
    assert torch.isfinite(flat).all()
    tol = GRAD_REL if dtype == 'float16' else 8e-2
    worst = {}
    for k in want:
        ref = leaves['visual.' + k].grad
        got = grads[k].reshape(ref.shape)
        worst[k] = (rel_l2(got, ref), cosine(got, ref))
    bad = {k: v for k, v in worst.items() if not (v[0] < tol and v[1] > (0.9995 if dtype == 'float16' else 0.995))}
    # the key bias has zero gradient in exact arithmetic (softmax does not see it): compare absolutely
    W = cfg['width']
    for k in list(bad):
        if k.endswith('attn.in_proj_bias'):
            ref = leaves['visual.' + k].grad
            got = grads[k].cpu().double()
            ok_qv = rel_l2(torch.cat([got[:W], got[2 * W:]]), torch.cat([ref[:W], ref[2 * W:]])) < tol
            ok_k = float(got[W:2 * W].abs().max()) < 1e-2 * float(ref.abs().max())
            if ok_qv and ok_k:
                bad.pop(k)
    assert not bad, bad


def test_gradients_at_the_full_depth_of_vit_l14(hip):
    """All 24 blocks of ViT-L/14 (the shipped fine-tuning configs' tower), two frames, against fp32 autograd of the
    oracle tower: the error of the 16-bit backward pass does not grow with depth beyond the tolerance."""
    from eventclip_amd import clip as eclip
    from oracle import clip_ref
    cfg = eclip.arch_config('ViT-L/14')
    model, tower, sd = _tower(cfg, seed=4)
    torch.manual_seed(7)
    n = 2
    imgs = torch.randn(n, 3, 224, 224)
    d_feats = torch.randn(n, cfg['embed_dim'])
    leaves = {k: v.float().clone().requires_grad_(True) for k, v in sd.items() if k.startswith('visual.')}
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    clip_ref.encode_image_autograd(leaves, cfg, imgs).backward(d_feats)
    names = list(tower.master)
    tower.forward(_patchify(tower, imgs))
    grads, flat = tower.backward(d_feats.cuda(), names)
    assert torch.isfinite(flat).all()
    worst = ('', 0.0)
    for k in names:
        if k.endswith('attn.in_proj_bias'):
            continue
        ref = leaves['visual.' + k].grad
        got = grads[k].reshape(ref.shape)
        e = rel_l2(got, ref)
        assert e < GRAD_REL and cosine(got, ref) > 0.9995, (k, e)
        worst = max(worst, (k, e), key=lambda t: t[1])
    print('worst relative l2 over', len(names), 'tensors:', worst)


def test_k_batched_tail_rows_of_a_reference_sized_batch(hip):
    """64 frames of ViT-L/14 are 64.25 row tiles: the K >= 2048 GEMMs send the last 64 rows through K-batches
    (csrc/vit_train.hip: gemm_rows32).  Features stay within fp32 reordering of the inference path, and the
    gradients of the batch equal the sum over its two halves (which take the single-launch path)."""
    cfg, _ = CONFIGS['l14_2blocks']
    model, tower, sd = _tower(cfg, seed=3)
    torch.manual_seed(8)
    n = 64
    imgs = torch.randn(n, 3, 224, 224)
    d_feats = torch.randn(n, cfg['embed_dim'], device='cuda')
    patches = _patchify(tower, imgs)
    feats = tower.forward(patches)
    want = model.encode_patches(patches)
    # (a different fp32 summation order in the tail rows moves a few 16-bit roundings downstream: the last frame's
    # features differ at the 1e-4 level, the others not at all)
    assert torch.equal(feats[:62], want[:62])
    torch.testing.assert_close(feats, want, rtol=0, atol=5e-4 * float(want.abs().max()))
    names = list(tower.master)
    whole, _ = tower.backward(d_feats, names)
    whole = {k: v.clone() for k, v in whole.items()}
    halves = {k: torch.zeros_like(v) for k, v in whole.items()}
    for lo in (0, 32):
        tower.forward(patches[lo:lo + 32].contiguous())
        part, _ = tower.backward(d_feats[lo:lo + 32].contiguous(), names)
        for k in names:
            halves[k] += part[k]
    for k in names:
        if k.endswith('attn.in_proj_bias'):
            continue                                   # zero key-bias gradient: see the tower test
        assert rel_l2(whole[k], halves[k]) < 2e-3, (k, rel_l2(whole[k], halves[k]))


@pytest.mark.parametrize('spec,name,dtype', [('qkvo-4', 'b32_2blocks', 'float16'), ('qkvo-16', 'l14_2blocks', 'float16'),
                                             ('qv-16', 'l14_2blocks', 'bfloat16'), ('qkvo-24', 'wide_odd', 'float16'),
                                             ('qkvo-64', 'b32_2blocks', 'bfloat16'), (40, 'tiny', 'float16'),
                                             ('qkv-3', 'l14_336_1block', 'float16')])
def test_lora_gradients_from_the_activations_follow_the_chain_rule(hip, spec, name, dtype):
    """The factor gradients the backward pass takes straight from the activations (csrc/vit_train.hip: lora_grads --
    projections written as transposed hi / lo planes, MFMA outer products over the row index, q / k / v sharing
    one read of ln_1(x) when r <= 16) against d up = dW down^T, d down = up^T dW with dW the merged weight's
    gradient from the same tower in full mode.  Ranks on both sides of one 16-row factor tile, both dtypes, row
    counts that are not multiples of 32."""
    from eventclip_amd import ft
    cfg, n = CONFIGS[name]
    model, tower, _ = _tower(cfg, seed=5, dtype=dtype)
    lf = ft.LoraFactors(tower, spec)
    torch.manual_seed(11)
    for k, p in lf.params.items():
        if 'lora_up' in k:
            p.copy_(torch.randn_like(p) * 0.02)
    grads = {k: torch.full_like(p, float('nan')) for k, p in lf.params.items()}
    lf.bind(grads)
    lf.merge()
    imgs = torch.randn(n, 3, cfg['image_size'], cfg['image_size'])
    d_feats = torch.randn(n, cfg['embed_dim'], device='cuda')
    patches = _patchify(tower, imgs)
    tower.forward(patches)
    tower.backward(d_feats, [], lf.struct)
    tower.forward(patches)
    full, _ = tower.backward(d_feats, list(lf.merged_names))
    W = tower.W
    tol = 3e-3 if dtype == 'float16' else 2e-2
    worst = 0.0
    for i, j, kd, ku in lf.projections():
        pre = f'transformer.resblocks.{i}.attn'
        dW = full[pre + ('.in_proj_weight' if j is not None else '.out_proj.weight')].view(-1, W)
        dW = dW[j * W:(j + 1) * W] if j is not None else dW
        # the kernels multiply by the 16-bit copies of the factors
        down, up = lf.down16[kd][:lf.r].float(), lf.up16t[ku][:lf.r].float().T
        for got, want in ((grads[ku], dW @ down.T), (grads[kd], up.T @ dW)):
            assert torch.isfinite(got).all()
            e = rel_l2(got, want)
            assert e < tol, (kd, e)
            worst = max(worst, e)
    print('worst relative l2:', worst)


@pytest.mark.parametrize('subset', ['all', 'bias', 'blocks_1_2'])
def test_staged_backward_is_the_single_pass_and_its_buckets_tile_the_gradient_buffer(hip, subset):
    cfg, n = CONFIGS['wide_odd']
    model, tower, sd = _tower(cfg, seed=2)
    torch.manual_seed(6)
    imgs = torch.randn(n, 3, cfg['image_size'], cfg['image_size'])
    d_feats = torch.randn(n, cfg['embed_dim'], device='cuda')
    patches = _patchify(tower, imgs)
    want = {'all': list(tower.master), 'bias': [k for k in tower.master if 'bias' in k],
            'blocks_1_2': [k for k in tower.master if '.1.' in k or '.2.' in k]}[subset]
    want = tower.canonical(want)
    tower.forward(patches)
    _, flat = tower.backward(d_feats, want)
    whole = flat.clone()
    plan = tower.bucket_plan(want, blocks_per_bucket=2)
    assert [p[:2] for p in plan] == [(0, 1), (1, 3), (3, 4), (4, 5)]          # head | blocks 2, 1 | block 0 | embedding
    covered = sorted((lo, hi) for _, _, lo, hi in plan if hi > lo)
    total = sum(tower.master[k].numel() for k in want)
    assert covered[0][0] == 0 and covered[-1][1] == total
    assert all(a[1] == b[0] for a, b in zip(covered, covered[1:]))            # contiguous, no overlap
    tower.forward(patches)
    flat.fill_(float('nan'))
    seen = torch.zeros_like(flat, dtype=torch.bool)
    for sb, se, lo, hi in plan:
        tower.backward(d_feats, want, stages=(sb, se))
        seen[lo:hi] = True
        # what this stage completed is final: equal to the single pass already
        assert torch.equal(flat[lo:hi], whole[lo:hi]), (sb, se)
    assert bool(seen[:total].all()) and torch.equal(flat[:total], whole[:total])


def test_skipping_gradients_does_not_change_the_ones_asked_for(hip):
    cfg, n = CONFIGS['wide_odd']
    model, tower, sd = _tower(cfg, seed=2)
    torch.manual_seed(6)
    imgs = torch.randn(n, 3, cfg['image_size'], cfg['image_size'])
    d_feats = torch.randn(n, cfg['embed_dim'], device='cuda')
    patches = _patchify(tower, imgs)
    tower.forward(patches)
    full, _ = tower.backward(d_feats, list(tower.master))
    full = {k: v.clone() for k, v in full.items()}
    subsets = (['proj'], ['ln_post.weight', 'ln_post.bias'],
               ['transformer.resblocks.2.attn.in_proj_weight', 'transformer.resblocks.1.attn.out_proj.weight'],
               [k for k in tower.master if 'bias' in k], ['class_embedding'], ['conv1.weight'],
               ['transformer.resblocks.1.mlp.c_fc.weight'])
    for want in subsets:
        tower.forward(patches)
        got, _ = tower.backward(d_feats, want)
        for k in want:
            assert torch.equal(got[k], full[k]), k


# ------------------------------------------------------------------------------------------------
# whole optimisation steps against the reference's classes
# ------------------------------------------------------------------------------------------------
def _golden_case(tag):
    from conftest import GOLDEN
    import test_oracle_ft_train as t
    z = np.load(os.path.join(GOLDEN, 'ft_train.npz'))
    return z, t.case(z, tag)


def _classifier_for(c, mixed_precision=True, init_scale=1024.0):
    from eventclip_amd import clip as eclip, ft
    from eventclip_amd.clip_cls_ft import FTCLIPClassifier
    cfg = _tiny_cfg(**c['cfg'])
    cfg.pop('heads', None)
    sd = eclip.random_state_dict(cfg, seed=0)
    plain = {k: v for k, v in c['visual'].items()}
    base = {}
    for k, v in plain.items():                     # the un-merged tower: merged_proj / linear.* back under plain names
        if '.lora_' in k:
            continue
        k2 = k.replace('.in_proj_weight.merged_proj', '.in_proj_weight').replace('.out_proj.linear.', '.out_proj.')
        base[k2] = v
    for k, v in base.items():
        sd['visual.' + k] = v.clone()
    model = eclip.CLIP(cfg, sd, full_last_block=True).cuda()
    K = c['text'].shape[0]
    cd = dict(clip_model=model, prompt='a point cloud image of a {}', class_names=[f'class_{i}' for i in range(K)],
              agg_func=c['agg'], class_tokens=eclip.synthetic_tokens(K), **c['clip_dict'])
    clf = FTCLIPClassifier(adapter_dict=dict(adapter_type='text-identity' if c['prompt'] else 'identity', residual=True),
                           clip_dict=cd, loss_dict=dict(use_logits_loss=not c['probs_loss'],
                                                        use_probs_loss=c['probs_loss'])).cuda()
    if c['prompt']:
        clf.text_feats.data.copy_(c['text'].cuda())
    else:
        clf.text_feats = c['text'].cuda().float()          # the fixed (already normalised) text features
        clf._text_t = None
    return clf


@pytest.mark.parametrize('tag', ['full', 'lora_qkvo', 'lora_int', 'lora_qv', 'bias', 'ln', 'conv_cls'])
def test_training_step_matches_the_reference(hip, tag):
    """Loss, every gradient and the parameters after two Adam steps, against the reference's FTCLIPClassifier
    under torch autograd + torch.optim.Adam (fp32)."""
    from eventclip_amd import ft
    z, c = _golden_case(tag)
    clf = _classifier_for(c)
    lr, clip_lr = float(z['lr']), float(z['clip_lr'])
    # constant learning rates (the golden run has no scheduler): total_steps huge, no warm-up
    tr = ft.FTTrainer(clf, lr=lr, clip_lr=clip_lr, total_steps=10 ** 9, warmup_steps_pct=0.0, init_scale=1024.0)
    if tr.lora:
        for k in tr.lora.params:
            tr.lora.params[k].copy_(c['visual'][k].cuda())
        tr.lora.merge()
    assert tr.trainable_names() == c['trainable']
    data = {'img': c['imgs'].cuda(), 'valid_mask': c['valid'].cuda(), 'label': c['labels'].cuda()}
    loss = tr.step(data)
    tr.resolve()
    assert not tr.last['skipped']
    assert abs(float(loss) - float(z[f'{tag}/loss'])) < 2e-3 * max(1.0, abs(float(z[f'{tag}/loss'])))
    np.testing.assert_allclose(tr.last['feats'].cpu().numpy(), z[f'{tag}/feats'], rtol=0,
                               atol=1e-3 * float(np.abs(z[f'{tag}/feats']).max()))
    for name in c['trainable']:
        want = torch.from_numpy(z[f'{tag}/grad:{name}'])
        got = tr.last['grads'][name].reshape(want.shape)
        if name.endswith('attn.in_proj_bias'):             # zero key-bias gradient: see the tower test
            W = c['cfg']['width']
            sel = torch.cat([torch.arange(W), torch.arange(2 * W, 3 * W)])
            assert rel_l2(got.cpu()[sel], want[sel]) < GRAD_REL, name
            continue
        assert rel_l2(got, want) < GRAD_REL and cosine(got, want) > 0.9995, (name, rel_l2(got, want))
    tr.step(data)
    sd = clf.state_dict()
    for name in c['trainable']:
        want = z[f'{tag}/step2:{name}']
        got = sd[name].detach().cpu().numpy().reshape(want.shape)
        g0 = np.abs(z[f'{tag}/grad:{name}'])
        live = g0 > 1e-2 * g0.max()        # Adam normalises the step: elements with ~zero gradient follow rounding
        np.testing.assert_allclose(got[live], want[live], rtol=0, atol=0.35 * clip_lr + 1e-6, err_msg=name)
    # the checkpoint carries the reference's key set for this configuration
    want_keys = set(z[f'{tag}/sd_keys'].tolist())
    assert {k for k in sd if k.startswith('model.visual.')} == {k for k in want_keys if k.startswith('model.visual.')}


def test_gradient_scaler_skips_and_backs_off(hip):
    from eventclip_amd import ft
    z, c = _golden_case('bias')
    clf = _classifier_for(c)
    tr = ft.FTTrainer(clf, lr=1e-3, clip_lr=1e-3, total_steps=100, init_scale=2.0 ** 40, growth_interval=2)
    data = {'img': c['imgs'].cuda(), 'valid_mask': c['valid'].cuda(), 'label': c['labels'].cuda()}
    before = {k: v.clone() for k, v in tr.tensors.items()}
    tr.step(data)                                   # 2^40 overflows the 16-bit gradients: skipped
    tr.resolve()
    assert tr.last['skipped'] and tr.scaler.scale == 2.0 ** 39 and tr.opt_steps == 0
    assert all(torch.equal(before[k], v) for k, v in tr.tensors.items())
    tr.scaler.scale = 256.0
    tr.step(data)
    tr.step(data)
    tr.resolve()
    assert tr.opt_steps == 2 and tr.scaler.scale == 512.0     # two clean steps: growth_interval = 2
    assert any(not torch.equal(before[k], v) for k, v in tr.tensors.items())


def test_graph_replay_takes_the_same_steps_as_eager_launches(hip):
    """graph=True records the GPU work of a step into a hipGraph and replays it; learning rates, Adam's bias
    corrections and the loss scale reach the kernels through device memory.  Five steps either way end at the same
    parameters (same kernels, same order: bit for bit)."""
    from eventclip_amd import ft
    z, c = _golden_case('lora_qkvo')
    results = []
    for graph in (False, True):
        clf = _classifier_for(c)
        torch.manual_seed(21)
        tr = ft.FTTrainer(clf, lr=1e-2, clip_lr=5e-3, total_steps=20, warmup_steps_pct=0.2, init_scale=512.0,
                          growth_interval=2, graph=graph, graph_warmup=2)
        t = tr.tower
        imgs = c['imgs'].cuda()[c['valid']]
        from eventclip_amd import _lib
        patches = torch.empty((imgs.shape[0], t.G, t.kpad), dtype=t.cd, device='cuda')
        _lib.check(_lib.lib().ec_patchify(_lib.ptr(imgs.float().contiguous()), imgs.shape[0], t.cfg['image_size'], t.P,
                                          t.kpad, _lib.ptr(patches), t.code, _lib.stream_ptr()), 'ec_patchify')
        valid = c['valid'].cuda()
        flat = valid.reshape(-1)
        row_idx = torch.where(flat, torch.cumsum(flat.int(), 0) - 1, torch.full_like(flat, -1, dtype=torch.int64))
        data = {'patches': patches, 'row_idx': row_idx.to(torch.int32).reshape(valid.shape), 'valid_mask': valid,
                'label': c['labels'].cuda()}
        losses = [float(tr.step(data)) for _ in range(5)]
        tr.resolve()
        assert (tr._graph is not None) == graph and tr.opt_steps == 5
        assert tr.scaler.scale == 512.0 * 4                     # two growths of two clean steps each
        results.append((losses, {k: v.clone() for k, v in tr.tensors.items()}))
    (l0, p0), (l1, p1) = results
    assert l0 == l1
    for k in p0:
        assert torch.equal(p0[k], p1[k]), k


def test_eval_between_steps_uses_the_trained_weights(hip):
    """forward() of the classifier mid-run encodes through the trainer's merged copies; a checkpoint written
    then and served by a fresh classifier gives the same logits."""
    from eventclip_amd import ft
    z, c = _golden_case('lora_qkvo')
    clf = _classifier_for(c)
    tr = ft.FTTrainer(clf, lr=1e-2, clip_lr=5e-3, total_steps=50, init_scale=1024.0)
    data = {'img': c['imgs'].cuda(), 'valid_mask': c['valid'].cuda(), 'label': c['labels'].cuda()}
    clf.eval()
    before = clf(data)['logits'].clone()
    clf.train()
    for _ in range(3):
        tr.step(data)
    clf.eval()
    after = clf(data)['logits']
    assert (after - before).abs().max() > 1e-3
    ckpt = {k: v.detach().cpu().clone() for k, v in clf.state_dict().items()}
    assert any('.lora_up_q' in k for k in ckpt)
    fresh = _classifier_for(c)
    fresh.load_state_dict(ckpt)
    fresh.eval()
    torch.testing.assert_close(fresh(data)['logits'], after, rtol=2e-3, atol=2e-3 * float(after.abs().max()))


@pytest.mark.parametrize('case', ['lora_qkvo', 'full'])
def test_resume_from_a_checkpoint_reproduces_the_next_step(hip, case):
    """The resume flow of the reference (nerv builds the optimiser, THEN loads the checkpoint): three steps, save
    model + trainer state, build a fresh classifier + trainer (different LoRA draw), load both, and the next step
    -- loss and every tensor after it -- equals the uninterrupted run's bit for bit.  With LoRA this only holds
    if the factors come back as factors over the frozen base (not folded into it) and Adam's moments, step counts
    and the loss scale travel too."""
    from eventclip_amd import ft
    z, c = _golden_case(case)
    data = {'img': c['imgs'].cuda(), 'valid_mask': c['valid'].cuda(), 'label': c['labels'].cuda()}
    kw = dict(lr=1e-2, clip_lr=5e-3, total_steps=20, warmup_steps_pct=0.2, init_scale=512.0, growth_interval=2)
    clf = _classifier_for(c)
    torch.manual_seed(21)
    tr = ft.FTTrainer(clf, **kw)
    for _ in range(3):
        tr.step(data)
    ckpt = {k: v.detach().cpu().clone() for k, v in clf.state_dict().items()}
    opt = {k: ({kk: vv.cpu().clone() for kk, vv in v.items()} if k.startswith('exp_avg') else v)
           for k, v in tr.state_dict().items()}
    want_loss = float(tr.step(data))
    tr.resolve()
    want = {k: v.clone() for k, v in tr.tensors.items()}
    clf.eval()
    want_logits = clf(data)['logits'].clone()

    clf2 = _classifier_for(c)
    torch.manual_seed(99)                                   # another draw of the LoRA factors: must not matter
    tr2 = ft.FTTrainer(clf2, **kw)
    clf2.load_state_dict(ckpt)
    tr2.load_state_dict(opt)
    assert tr2.steps == 3 and tr2.opt_steps == 3 and tr2.scaler.scale == opt['scaler']['scale']
    got_loss = float(tr2.step(data))
    tr2.resolve()
    assert got_loss == want_loss
    for k in want:
        assert torch.equal(tr2.tensors[k], want[k]), k
    clf2.eval()
    assert torch.equal(clf2(data)['logits'], want_logits)   # eval after the load runs on the trained weights


def test_two_ranks_with_different_seeds_start_from_rank_zero(hip):
    """FTTrainer under torch.distributed broadcasts rank 0's tensors at construction (DistributedDataParallel's
    behaviour): two ranks that seed their LoRA draw differently hold identical factors before and after steps."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EVENTCLIP_DIST_BACKEND='gloo', MASTER_ADDR='127.0.0.1', FT_DDP_SEED_PER_RANK='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', '29549', 'tools/ft_ddp_check.py', 'lora']
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert d['seed_per_rank'] and d['params_equal_at_start'] and d['params_equal_after_steps'], d


@pytest.mark.parametrize('mode', ['full', 'lora'])
def test_two_ranks_average_their_gradients(hip, mode):
    """FTTrainer under torch.distributed (two gloo ranks sharing the GPU): one all-reduce of the flat gradient
    buffer per step makes the ranks' step equal to a single process stepping on the whole batch."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EVENTCLIP_DIST_BACKEND='gloo', MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', '29547', 'tools/ft_ddp_check.py', mode]
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert d['world'] == 2 and not d['skipped'] and d['tensors'] > 10
    assert abs(d['loss_mean_of_ranks'] - d['loss_whole']) < 1e-3 * max(1.0, abs(d['loss_whole']))
    # the ranks' averaged gradients are the whole batch's (16-bit operands: not bit for bit)
    assert d['worst_grad_rel_l2'] < 1e-2, d


@pytest.mark.parametrize('mode', ['lora', 'full'])
def test_fine_tuning_learns_a_separable_toy_problem(hip, mode):
    """Forty steps of FTTrainer (hipGraph replay) on frames whose class is a spatial pattern the random tower does
    not separate: the loss falls and the training batch ends up classified -- every piece of the step (tape,
    backward, LoRA / full gradients, scaler, Adam, re-packing) pulls in the same direction."""
    from eventclip_amd import _lib, clip as eclip, ft
    from eventclip_amd.clip_cls_ft import FTCLIPClassifier
    cfg = _tiny_cfg(image_size=48, patch=16, width=128, layers=2, embed_dim=32)
    model = eclip.CLIP(cfg, eclip.random_state_dict(cfg, seed=0), full_last_block=True).cuda()
    K, B, T = 4, 16, 2
    cd = dict(clip_model=model, prompt='a point cloud image of a {}', class_names=[f'c{i}' for i in range(K)],
              agg_func='mean', class_tokens=eclip.synthetic_tokens(K), only_conv1=False, only_bias=False, only_ln=False,
              lora='qkvo-4' if mode == 'lora' else -1)
    clf = FTCLIPClassifier(adapter_dict=dict(adapter_type='text-identity', residual=True), clip_dict=cd,
                           loss_dict=dict(use_logits_loss=True, use_probs_loss=False)).cuda().train()
    torch.manual_seed(1)
    tr = ft.FTTrainer(clf, lr=5e-3, clip_lr=2e-3, total_steps=40, warmup_steps_pct=0.1, init_scale=1024.0, graph=True)
    t = tr.tower
    g = torch.Generator().manual_seed(2)
    labels = torch.arange(B) % K
    imgs = 0.3 * torch.randn(B, T, 3, 48, 48, generator=g)
    for b in range(B):                                 # class = which quadrant carries a bright blob
        y0, x0 = (labels[b] // 2) * 24, (labels[b] % 2) * 24
        imgs[b, :, :, y0:y0 + 24, x0:x0 + 24] += 1.5
    flat_imgs = imgs.reshape(B * T, 3, 48, 48).cuda().contiguous()
    patches = torch.empty((B * T, t.G, t.kpad), dtype=t.cd, device='cuda')
    _lib.check(_lib.lib().ec_patchify(_lib.ptr(flat_imgs), B * T, 48, t.P, t.kpad, _lib.ptr(patches), t.code,
                                      _lib.stream_ptr()), 'ec_patchify')
    data = {'patches': patches, 'row_idx': torch.arange(B * T, dtype=torch.int32, device='cuda').view(B, T),
            'valid_mask': torch.ones(B, T, dtype=torch.bool, device='cuda'), 'label': labels.cuda()}
    losses = [float(tr.step(data)) for _ in range(40)]
    tr.resolve()
    assert tr._graph is not None and tr.opt_steps >= 38
    assert losses[-1] < 0.25 * losses[0], (losses[0], losses[-1])
    clf.eval()
    pred = clf(data)['logits'].argmax(-1).cpu()
    assert (pred == labels).float().mean() >= 0.9, pred.tolist()
