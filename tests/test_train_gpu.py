"""Few-shot text-identity training step on the MI355X against the reference's autograd vectors, the
float64 oracle at full size, and torch.optim.Adam."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def golden_cases():
    z = np.load(os.path.join(GOLDEN, 'train_text_identity.npz'))
    return [(ci, agg, loss) for ci in range(len(z['cases'])) for agg in ('sum', 'mean') for loss in ('logits', 'probs')]


@pytest.mark.parametrize('ci,agg,loss', golden_cases())
def test_loss_and_gradient_match_reference_autograd(ci, agg, loss, hip):
    import torch
    from eventclip_amd import train
    z = np.load(os.path.join(GOLDEN, 'train_text_identity.npz'))
    tag = f'c{ci}_{agg}_{loss}'
    f = torch.from_numpy(z[f'c{ci}_feats']).cuda()
    v = torch.from_numpy(z[f'c{ci}_valid']).cuda()
    y = torch.from_numpy(z[f'c{ci}_labels']).cuda()
    t = torch.from_numpy(z[f'c{ci}_text_param']).cuda()
    got_loss, got_grad, got_logits = train.fs_text_loss_grad(f, v, y, t, float(z[f'c{ci}_logit_scale']), agg,
                                                             loss == 'probs', return_logits=True)
    # fp32 on both sides (torch CPU autograd vs HIP): different summation orders
    want_loss, g = float(z[tag + '_loss']), z[tag + '_grad']
    assert abs(float(got_loss) - want_loss) < 1e-4 * max(1., abs(want_loss))
    assert np.abs(got_grad.cpu().numpy() - g).max() < 1e-4 * max(np.abs(g).max(), 1e-3)
    np.testing.assert_allclose(got_logits.cpu().numpy(), z[tag + '_logits'], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('B,T,D,K,agg,probs', [(64, 10, 768, 101, 'mean', True), (32, 1, 768, 2, 'sum', False),
                                                (16, 5, 768, 1000, 'mean', False), (7, 3, 512, 100, 'sum', True)])
def test_full_size_against_oracle(B, T, D, K, agg, probs, hip):
    import torch
    from eventclip_amd import train
    from oracle import train as ot
    rng = np.random.default_rng(B + K)
    feats = rng.standard_normal((B, T, D)).astype(np.float32)
    valid = rng.random((B, T)) < 0.7
    valid[:, 0] = True
    labels = rng.integers(0, K, B)
    text = (rng.standard_normal((K, D)) * 0.5).astype(np.float32)
    # few-shot regime: views correlate with their class row so the loss is not just log K
    feats += 2.0 * text[labels][:, None, :]
    want_loss, want_grad, want_logits = ot.fs_text_loss_and_grad(feats, valid, labels, text, 100.0, agg, probs)
    got_loss, got_grad, got_logits = train.fs_text_loss_grad(
        torch.from_numpy(feats).cuda(), torch.from_numpy(valid).cuda(), torch.from_numpy(labels).cuda(),
        torch.from_numpy(text).cuda(), 100.0, agg, probs, return_logits=True)
    assert abs(float(got_loss) - want_loss) < 2e-4 * max(1., abs(want_loss))
    np.testing.assert_allclose(got_logits.cpu().numpy(), want_logits, rtol=2e-4, atol=2e-3)
    assert np.abs(got_grad.cpu().numpy() - want_grad).max() < 3e-4 * np.abs(want_grad).max()


def test_adam_kernel_matches_torch_adam(hip):
    import torch
    from eventclip_amd import train
    g = torch.Generator(device='cuda').manual_seed(3)
    p0 = torch.randn(101, 768, device='cuda', generator=g)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=2e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.02)
    p, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    for step in range(1, 8):
        grad = torch.randn(101, 768, device='cuda', generator=g) * (0.1 if step % 2 else 3.0)
        ref.grad = grad.clone()
        opt.step()
        train.adam_step(p, grad, m, v, step, 2e-3, (0.9, 0.98), 1e-8, 0.02)
        torch.testing.assert_close(p, ref.data, rtol=2e-5, atol=2e-6)


def test_trainer_fits_a_few_shot_problem(hip):
    """End to end on cached features of the real classifier: loss falls, accuracy rises to 100 %."""
    import torch
    from eventclip_amd import clip as eclip
    from eventclip_amd.clip_cls import FSCLIPClassifier
    from eventclip_amd.train import TextFeatTrainer
    cfg = eclip.arch_config('ViT-B/32', layers=1, text_layers=1, vocab_size=49408)
    m = eclip.CLIP(cfg, eclip.random_state_dict(cfg, seed=9)).cuda().eval()
    K, B, T = 5, 20, 3
    clf = FSCLIPClassifier(adapter_dict=dict(adapter_type='text-identity', in_dim=cfg['embed_dim'], residual=True),
                           clip_dict=dict(clip_model=m, prompt='a {}', class_names=[f'c{i}' for i in range(K)],
                                          agg_func='mean', class_tokens=eclip.synthetic_tokens(K, seed=4)),
                           loss_dict=dict(use_logits_loss=False, use_probs_loss=True)).cuda()
    g = torch.Generator(device='cuda').manual_seed(1)
    protos = torch.randn(K, 3, 224, 224, device='cuda', generator=g)
    labels = torch.arange(B, device='cuda') % K
    imgs = protos[labels][:, None] + 0.3 * torch.randn(B, T, 3, 224, 224, device='cuda', generator=g)
    valid = torch.ones(B, T, dtype=torch.bool, device='cuda')
    valid[::4, 2] = False
    feats, vm = clf.cache_feats({'img': imgs, 'valid_mask': valid})
    assert torch.equal(vm, valid) and feats.shape == (B, T, cfg['embed_dim']) and float(feats[0, 2].abs().max()) == 0.
    # plumbing on the encoder's own (random-weight, nearly class-blind) features: a step runs and moves the prompts
    before = clf.text_feats.data.clone()
    loss0 = float(TextFeatTrainer(clf, lr=1e-3, total_steps=10).step(feats, valid, labels))
    assert np.isfinite(loss0) and not torch.equal(before, clf.text_feats.data)
    # convergence on separable cached features (what a trained encoder provides)
    D = cfg['embed_dim']
    centres = torch.randn(K, D, device='cuda', generator=g)
    sep = centres[labels][:, None] + 0.4 * torch.randn(B, T, D, device='cuda', generator=g)
    trainer = TextFeatTrainer(clf, lr=1e-2, total_steps=80, warmup_steps_pct=0.05)
    losses = [float(trainer.step(sep, valid, labels)) for _ in range(80)]
    assert losses[-1] < 0.1 * losses[0] and losses[-1] < 0.05
    _, _, logits = __import__('eventclip_amd.train', fromlist=['x']).fs_text_loss_grad(
        sep, valid, labels, clf.text_feats.data, clf.logit_scale, clf.agg_func, clf.use_probs_loss, return_logits=True)
    assert float((logits.argmax(-1) == labels).float().mean()) == 1.0


def _trans_adapter(z, dev='cuda'):
    import torch
    from eventclip_amd.adapter import TransformerAdapter
    ad = TransformerAdapter(in_dim=int(z['C']), d_model=int(z['adcfg_d_model']), num_heads=int(z['adcfg_num_heads']),
                            ffn_dim=int(z['adcfg_ffn_dim']), num_layers=int(z['adcfg_num_layers']),
                            residual=float(z['adcfg_residual']))
    ad.load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('w:')})
    return ad.to(dev)


@pytest.mark.parametrize('ci,agg,loss', [(ci, a, l) for ci in range(3) for a, l in (('sum', 'logits'), ('mean', 'probs'))])
def test_text_trans_gradients_match_reference_autograd(ci, agg, loss, hip):
    import torch
    from eventclip_amd import train
    z = np.load(os.path.join(GOLDEN, 'train_text_trans.npz'))
    tag = f'c{ci}_{agg}_{loss}'
    ad = _trans_adapter(z)
    f = torch.from_numpy(z[f'c{ci}_feats']).cuda()
    v = torch.from_numpy(z[f'c{ci}_valid']).cuda()
    y = torch.from_numpy(z[f'c{ci}_labels']).cuda()
    t = torch.from_numpy(z[f'c{ci}_text_param']).cuda()
    got_loss, grads, logits = train.fs_trans_loss_grad(f, v, y, t, float(z[f'c{ci}_logit_scale']), ad, agg,
                                                       loss == 'probs', return_logits=True)
    want_loss = float(z[tag + '_loss'])
    assert abs(float(got_loss) - want_loss) < 2e-4 * max(1., abs(want_loss))
    np.testing.assert_allclose(logits.cpu().numpy(), z[tag + '_logits'], rtol=2e-4, atol=2e-3)
    assert set(grads) == {k.split('_g:')[1] for k in z.files if k.startswith(tag + '_g:')}
    for k, g in grads.items():
        want = z[f'{tag}_g:{k}']
        err = np.abs(g.cpu().numpy() - want).max()
        assert err < 5e-4 * max(np.abs(want).max(), 1e-3), (k, err, np.abs(want).max())


def test_text_trans_full_size_against_oracle(hip):
    """The shipped adapter geometry (in 768, d_model 256, 4 heads, ffn 1024, 2 layers, r = 0.95), T = 10."""
    import torch
    from eventclip_amd import train
    from eventclip_amd.adapter import TransformerAdapter
    from oracle import train as ot
    torch.manual_seed(3)
    B, T, D, K = 24, 10, 768, 101
    ad = TransformerAdapter(in_dim=D, residual=0.95)
    with torch.no_grad():
        for p in ad.parameters():
            p.add_(torch.randn_like(p) * 0.02)
    valid = torch.rand(B, T) < 0.7
    valid[:, 0] = True
    labels = torch.randint(0, K, (B,))
    text = torch.randn(K, D) * 0.5
    feats = (torch.randn(B, T, D) + 2.0 * text[labels][:, None]) * valid[..., None]
    want_loss, want, want_logits = ot.fs_trans_loss_and_grads(
        {k: v.detach().numpy() for k, v in ad.state_dict().items()}, feats.numpy(), valid.numpy(), labels.numpy(),
        text.numpy(), 100.0, 4, 0.95, 'mean', True)
    got_loss, grads = train.fs_trans_loss_grad(feats.cuda(), valid.cuda(), labels.cuda(), text.cuda(), 100.0,
                                               ad.cuda(), 'mean', True)
    assert abs(float(got_loss) - want_loss) < 3e-4 * max(1., abs(want_loss))
    for k, g in grads.items():
        err = np.abs(g.cpu().numpy() - want[k]).max()
        assert err < 1e-3 * max(np.abs(want[k]).max(), 1e-4), (k, err, np.abs(want[k]).max())


def test_adapter_trainer_reduces_the_loss_and_updates_the_forward(hip):
    import torch
    from eventclip_amd import clip as eclip
    from eventclip_amd.clip_cls import FSCLIPClassifier
    from eventclip_amd.train import AdapterTrainer
    cfg = eclip.arch_config('ViT-B/32', layers=1, text_layers=1, vocab_size=49408)
    m = eclip.CLIP(cfg, eclip.random_state_dict(cfg, seed=9)).cuda().eval()
    K, B, T, D = 4, 16, 3, cfg['embed_dim']
    clf = FSCLIPClassifier(adapter_dict=dict(adapter_type='text-trans', in_dim=D, d_model=64, num_heads=2, ffn_dim=128,
                                             num_layers=2, residual=0.5),
                           clip_dict=dict(clip_model=m, prompt='a {}', class_names=[f'c{i}' for i in range(K)],
                                          agg_func='mean', class_tokens=eclip.synthetic_tokens(K, seed=4)),
                           loss_dict=dict(use_logits_loss=True, use_probs_loss=False)).cuda()
    g = torch.Generator(device='cuda').manual_seed(2)
    labels = torch.arange(B, device='cuda') % K
    centres = torch.randn(K, D, device='cuda', generator=g)
    valid = torch.ones(B, T, dtype=torch.bool, device='cuda')
    valid[1::3, 2] = False
    feats = (centres[labels][:, None] + 0.5 * torch.randn(B, T, D, device='cuda', generator=g)) * valid[..., None]
    w0 = clf.adapter.in_proj.weight.data.clone()
    trainer = AdapterTrainer(clf, lr=3e-3, total_steps=60)
    losses = [float(trainer.step(feats, valid, labels)) for _ in range(60)]
    assert losses[-1] < 0.2 * losses[0]
    assert not torch.equal(w0, clf.adapter.in_proj.weight.data)
    # the inference kernel sees the trained weights: same aggregated logits as the training forward
    from eventclip_amd.train import fs_trans_loss_grad
    _, _, logits = fs_trans_loss_grad(feats, valid, labels, clf.text_feats.data, clf.logit_scale, clf.adapter,
                                      clf.agg_func, clf.use_probs_loss, return_logits=True)
    idx = torch.where(valid, torch.arange(B * T, device='cuda').view(B, T), torch.full((B, T), -1, device='cuda')).int()
    ad = clf.adapter.forward_rows(feats.reshape(B * T, D), idx)
    fn = torch.nn.functional.normalize(ad, dim=-1) * valid[..., None]
    tx = torch.nn.functional.normalize(clf.text_feats.data, dim=-1)
    want = (clf.logit_scale * fn @ tx.T).sum(1) / valid.sum(1, keepdim=True)
    torch.testing.assert_close(logits, want, rtol=1e-3, atol=1e-2)
    assert float((logits.argmax(-1) == labels).float().mean()) == 1.0


def test_text_trans_dropout_replayed_in_the_oracle(hip):
    """Train-mode dropout: the kernels' masks (ec_dropout_mask) replayed in the float64 oracle give the
    same loss and gradients; the masks have the right rate and depend on seed and site."""
    import torch
    from eventclip_amd import train
    from eventclip_amd.adapter import TransformerAdapter
    from oracle import train as ot
    torch.manual_seed(5)
    B, T, D, K, d, ffn, heads, p_drop, seed = 12, 6, 96, 9, 64, 128, 4, 0.3, 987654321
    ad = TransformerAdapter(in_dim=D, d_model=d, num_heads=heads, ffn_dim=ffn, num_layers=2, residual=0.6)
    with torch.no_grad():
        for q in ad.parameters():
            q.add_(torch.randn_like(q) * 0.05)
    valid = torch.rand(B, T) < 0.7
    valid[:, 0] = True
    labels = torch.randint(0, K, (B,))
    text = torch.randn(K, D) * 0.5
    feats = (torch.randn(B, T, D) + 0.15 * text[labels][:, None]) * valid[..., None]   # hard: the loss stays > 0
    sizes = {0: B * heads * T * T, 1: B * T * d, 2: B * T * ffn, 3: B * T * d}
    masks = {(l, s): train.dropout_mask(seed, 4 * l + s, n, p_drop).cpu().numpy() for l in range(2) for s, n in sizes.items()}
    for (l, s_), mk in masks.items():
        assert abs(mk.mean() - (1 - p_drop)) < 0.05, (l, s_, mk.mean())
    assert not np.array_equal(masks[(0, 1)], masks[(0, 3)]) and not np.array_equal(masks[(0, 1)], masks[(1, 1)])
    assert not np.array_equal(masks[(0, 1)], train.dropout_mask(seed + 1, 1, sizes[1], p_drop).cpu().numpy())
    want_loss, want, _ = ot.fs_trans_loss_and_grads(
        {k: v.detach().numpy() for k, v in ad.state_dict().items()}, feats.numpy(), valid.numpy(), labels.numpy(),
        text.numpy(), 100.0, heads, 0.6, 'mean', False, dropout_p=p_drop, masks=masks)
    got_loss, grads = train.fs_trans_loss_grad(feats.cuda(), valid.cuda(), labels.cuda(), text.cuda(), 100.0,
                                               ad.cuda(), 'mean', False, dropout_p=p_drop, seed=seed)
    assert abs(float(got_loss) - want_loss) < 3e-4 * max(1., abs(want_loss))
    for k, g in grads.items():
        err = np.abs(g.cpu().numpy() - want[k]).max()
        assert err < 1e-3 * max(np.abs(want[k]).max(), 1e-4), (k, err, np.abs(want[k]).max())
    # and it is not the deterministic function
    det_loss, _ = train.fs_trans_loss_grad(feats.cuda(), valid.cuda(), labels.cuda(), text.cuda(), 100.0, ad, 'mean', False)
    assert want_loss > 0.1 and abs(float(det_loss) - float(got_loss)) > 1e-3


def test_two_rank_training_equals_one_rank_full_batch(hip, tmp_path):
    """DDP semantics of AdapterTrainer: two ranks (gloo, sharing the one GPU) on half batches with the
    gradient all-reduce reach the same parameters as one rank on the full batch (mean losses, equal shards)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    worker = os.path.join(root, 'tests', 'ddp_train_worker.py')
    one, two = str(tmp_path / 'one.npz'), str(tmp_path / 'two.npz')
    r = subprocess.run([sys.executable, worker, one], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', '29544', worker, two], cwd=root,
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    a, b = np.load(one), np.load(two)
    assert set(a.files) == set(b.files) and len(a.files) == 29
    for k in a.files:
        np.testing.assert_allclose(b[k], a[k], rtol=2e-4, atol=2e-6, err_msg=k)
