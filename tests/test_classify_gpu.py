"""ec_classify against the clip_cls.py restatement (MI355X)."""
import pytest

pytestmark = pytest.mark.gpu


def run_classify(feats, row_idx, text, scale, agg, normalize):
    import torch
    from eventclip_amd import _lib
    B, T = row_idx.shape
    K, C = text.shape
    full = torch.empty(B, T, K, device='cuda')
    logits = torch.empty(B, K, device='cuda')
    probs = torch.empty(B, K, device='cuda')
    text_t = text.t().contiguous()
    code = {'sum': _lib.EC_AGG_SUM, 'mean': _lib.EC_AGG_MEAN, 'max': _lib.EC_AGG_MAX}[agg]
    n_rows = feats.shape[0]
    tws = torch.empty(_lib.lib().ec_classify_text_bytes(C, K), dtype=torch.uint8, device='cuda')
    _lib.check(_lib.lib().ec_classify_prep_text(_lib.ptr(text_t), C, K, _lib.ptr(tws), tws.numel(), _lib.stream_ptr()))
    ws = torch.empty(max(_lib.lib().ec_classify_v2_workspace_bytes(n_rows, C, K), 256), dtype=torch.uint8, device='cuda')
    _lib.check(_lib.lib().ec_classify_v2(_lib.ptr(feats), n_rows, _lib.ptr(row_idx), _lib.ptr(tws), B, T, C,
                                         K, scale, code, int(normalize), _lib.ptr(full),
                                         _lib.ptr(logits), _lib.ptr(probs), _lib.ptr(ws), ws.numel(), _lib.stream_ptr()))
    return full, logits, probs


def check(full, logits, probs, want, T):
    """fp32 on both sides; only the summation order differs, so the tolerance is a few
    ulp of the largest logit (x T for the view sums)."""
    import torch
    mag = float(want['full_logits'].abs().max())
    torch.testing.assert_close(full.cpu(), want['full_logits'], rtol=1e-5, atol=2e-6 * mag)
    torch.testing.assert_close(logits.cpu(), want['logits'], rtol=1e-5, atol=2e-6 * mag * T)
    torch.testing.assert_close(probs.cpu(), want['probs'], rtol=2e-4, atol=1e-6)


@pytest.mark.parametrize('agg', ['sum', 'mean', 'max'])
@pytest.mark.parametrize('B,T,C,K', [(6, 10, 768, 101), (4, 1, 768, 2), (3, 5, 512, 1000),
                                     (5, 2, 64, 7)])
def test_zero_shot_matches_oracle(B, T, C, K, agg, hip):
    import torch
    import torch.nn.functional as F
    from oracle import classify as oc
    g = torch.Generator().manual_seed(B * 100 + T)
    valid = torch.rand(B, T, generator=g) < 0.7
    valid[:, 0] = True
    nv = int(valid.sum())
    feats = torch.randn(nv, C, generator=g) * 0.5
    text = F.normalize(torch.randn(K, C, generator=g), dim=-1)
    want = oc.zs_forward(feats, valid, text, 100.0, agg)
    idx = torch.full((B, T), -1, dtype=torch.int32)
    idx[valid] = torch.arange(nv, dtype=torch.int32)
    full, logits, probs = run_classify(feats.cuda(), idx.cuda(), text.cuda(), 100.0, agg, False)
    check(full, logits, probs, want, T)


@pytest.mark.parametrize('agg', ['sum', 'mean', 'max'])
def test_few_shot_tail_matches_oracle(agg, hip):
    import torch
    import torch.nn.functional as F
    from oracle import classify as oc
    g = torch.Generator().manual_seed(3)
    B, T, C, K = 5, 5, 768, 100
    valid = torch.rand(B, T, generator=g) < 0.6
    valid[:, 0] = True
    feats = torch.randn(B, T, C, generator=g)
    text = F.normalize(torch.randn(K, C, generator=g), dim=-1)
    want = oc.fs_tail(feats, valid, text, 100.0, agg)
    idx = torch.where(valid, torch.arange(B * T).view(B, T), torch.tensor(-1)).to(torch.int32)
    full, logits, probs = run_classify(feats.view(B * T, C).cuda(), idx.cuda(), text.cuda(), 100.0,
                                       agg, True)
    check(full, logits, probs, want, T)


def test_unnormalised_text_and_tiny_features(hip):
    """Nothing in the product depends on the text rows being unit vectors or on the features' magnitude: every feature row and
    every class row is scaled by its own power of two before the fp16 split and the scales are taken out again exactly."""
    import torch
    g = torch.Generator().manual_seed(2)
    B, T, C, K = 3, 2, 128, 20
    feats = torch.randn(B * T, C, generator=g) * torch.tensor([1e-4, 1.0, 300.0, 1e-2, 5.0, 1e3])[:, None]
    text = torch.randn(K, C, generator=g) * (10.0 ** torch.linspace(-3, 3, K))[:, None]
    idx = torch.arange(B * T, dtype=torch.int32).view(B, T)
    full, logits, probs = run_classify(feats.cuda(), idx.cuda(), text.cuda(), 1.0, 'sum', False)
    want = (feats.double() @ text.double().t()).view(B, T, K)
    scale = feats.double().abs().max(1).values.view(B, T, 1) * text.double().abs().max(1).values.view(1, 1, K) * C ** 0.5
    assert float(((full.cpu().double() - want).abs() / scale).max()) < 2e-6


def test_empty_and_ragged_edges(hip):
    """The edges the reference reaches (clip_cls.py:139: `imgs[valid_masks]` may be empty for a sample, never for a batch;
    an empty batch never reaches forward): a sample without a valid view gives zero logits rows / NaN-free probabilities
    semantics of its own (sum over nothing: 0, mean: 0 / 0), B = 0 is a no-op, and a feature matrix with more rows than
    row_idx uses (the full, un-compacted form) is indexed, not assumed compact."""
    import torch
    from eventclip_amd import _lib
    lib = _lib.lib()
    g = torch.Generator().manual_seed(1)
    C, K, T = 64, 7, 3
    feats = torch.randn(10, C, generator=g).cuda()
    text = torch.nn.functional.normalize(torch.randn(K, C, generator=g), dim=-1).cuda()
    idx = torch.tensor([[7, -1, 2], [-1, -1, -1]], dtype=torch.int32).cuda()          # rows 7 and 2 of a 10-row matrix; an empty sample
    full, logits, probs = run_classify(feats, idx, text, 100.0, 'sum', False)
    want0 = 100.0 * feats[[7, 2]] @ text.t()
    torch.testing.assert_close(full[0, [0, 2]], want0, rtol=1e-5, atol=1e-4)
    assert float(full[0, 1].abs().max()) == 0 and float(full[1].abs().max()) == 0
    torch.testing.assert_close(logits[0], want0.sum(0), rtol=1e-5, atol=1e-3)
    assert float(logits[1].abs().max()) == 0
    # B = 0: nothing is launched, nothing is touched
    e = torch.empty(0, device='cuda')
    ws = torch.empty(256, dtype=torch.uint8, device='cuda')
    tws = torch.empty(lib.ec_classify_text_bytes(C, K), dtype=torch.uint8, device='cuda')
    text_t = text.t().contiguous()
    assert lib.ec_classify_prep_text(_lib.ptr(text_t), C, K, _lib.ptr(tws), tws.numel(), _lib.stream_ptr()) == 0
    assert lib.ec_classify_v2(_lib.ptr(feats), 10, _lib.ptr(idx), _lib.ptr(tws), 0, T, C, K, 100.0, 0, 0, _lib.ptr(e),
                              _lib.ptr(e), _lib.ptr(e), _lib.ptr(ws), 256, _lib.stream_ptr()) == 0
    # a workspace that is too small is refused with the size that is needed -- for the call and for the text planes
    rc = lib.ec_classify_v2(_lib.ptr(feats), 10, _lib.ptr(idx), _lib.ptr(tws), 2, T, C, K, 100.0, 0, 0, _lib.ptr(full),
                            _lib.ptr(logits), _lib.ptr(probs), _lib.ptr(ws), 256, _lib.stream_ptr())
    assert rc != 0 and b'ec_classify_v2_workspace_bytes' in lib.ec_last_error()
    rc = lib.ec_classify_prep_text(_lib.ptr(text_t), C, K, _lib.ptr(tws), 256, _lib.stream_ptr())
    assert rc != 0 and b'ec_classify_text_bytes' in lib.ec_last_error()
    # the entry point of rounds 1 - 5 answers with an error code whatever it is handed (ABI 600)
    assert lib.ec_classify() == _lib.EC_ERR_UNSUPPORTED and b'ec_classify_v2' in lib.ec_last_error()


def test_text_planes_are_prepared_once_per_text_tensor():
    """torch.ops.eventclip_hip.classify builds the transposed hi + lo text planes once per text_t tensor (identity and
    version): a second batch against the same text reuses them, an in-place update or a new tensor rebuilds them."""
    import torch
    from eventclip_amd import torch_ops
    g = torch.Generator().manual_seed(3)
    C, K = 96, 37
    feats = torch.randn(6, C, generator=g).cuda()
    idx = torch.arange(6, dtype=torch.int32).view(3, 2).cuda()
    text_t = torch.nn.functional.normalize(torch.randn(K, C, generator=g), dim=-1).t().contiguous().cuda()
    torch_ops._TEXT_PLANES.clear()
    a = torch.ops.eventclip_hip.classify(feats, idx, text_t, 100.0, 1, False)
    assert len(torch_ops._TEXT_PLANES) == 1
    planes = next(iter(torch_ops._TEXT_PLANES.values()))[0]
    b = torch.ops.eventclip_hip.classify(feats, idx, text_t, 100.0, 1, False)
    assert len(torch_ops._TEXT_PLANES) == 1 and next(iter(torch_ops._TEXT_PLANES.values()))[0] is planes
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    text_t.mul_(2.0)                                             # in-place: new version, planes rebuilt
    c = torch.ops.eventclip_hip.classify(feats, idx, text_t, 100.0, 1, False)
    torch.testing.assert_close(c[0], 2 * a[0], rtol=1e-6, atol=1e-4)
    want = 100.0 * feats.double() @ text_t.double()
    torch.testing.assert_close(c[0].reshape(6, K).double(), want, rtol=1e-5, atol=1e-4)
