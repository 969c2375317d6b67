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
    ws = torch.empty(_lib.lib().ec_classify_workspace_bytes(n_rows, C, K), dtype=torch.uint8, device='cuda')
    _lib.check(_lib.lib().ec_classify(_lib.ptr(feats), n_rows, _lib.ptr(row_idx), _lib.ptr(text_t), B, T, C,
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
