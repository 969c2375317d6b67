"""Time the few-shot training steps on cached encoder features at the reference's batch
shapes: `text-identity` (configs/fsclip/text_adapter: only text_feats trains) and `text-trans`
(configs/fsclip/joint_adapter: TransformerAdapter d_model 256, 2 layers, 4 heads + text_feats), train_batch_size 128,
views T, D = 768.

    python tools/bench_fs_train.py [--batch 128] [--steps 50]

One JSON line per (adapter type, geometry): ms per step and samples/s, and `cpu_baseline`: the oracle's step (loss +
gradients in float64 -- numpy for `text-identity`, torch CPU autograd over the explicit forward for `text-trans` --
plus its numpy Adam over every trained tensor) on the same batch, on the box's host cores.  Synthetic features,
seeded weights.  The CPU leg is the only place the oracle is touched."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def cpu_step(kind, clf, feats, valid, labels, repeat=3):
    from oracle import train as ot
    f, v, y = feats.cpu().numpy(), valid.cpu().numpy(), labels.cpu().numpy()
    text = clf.text_feats.detach().cpu().numpy().astype(np.float64)
    sd = {k: t.detach().cpu().numpy() for k, t in clf.adapter.state_dict().items()} if kind == 'text-trans' else {}
    params = dict({k: a.astype(np.float64) for k, a in sd.items()}, text_feats=text)
    m1 = {k: np.zeros_like(a) for k, a in params.items()}
    m2 = {k: np.zeros_like(a) for k, a in params.items()}
    best = float('inf')
    for it in range(repeat):
        t0 = time.perf_counter()
        if kind == 'text-identity':
            _, g, _ = ot.fs_text_loss_and_grad(f, v, y, params['text_feats'], 100.0, 'mean', False)
            grads = dict(text_feats=g)
        else:
            _, grads, _ = ot.fs_trans_loss_and_grads({k: params[k] for k in sd}, f, v, y, params['text_feats'], 100.0, 4, 0.95,
                                                     'mean', False)
        for k in params:
            ot.adam_step(params[k], grads[k], m1[k], m2[k], it + 1, 2e-5)
        best = min(best, time.perf_counter() - t0)
    return dict(ms_per_step=round(best * 1e3, 2), kind='port', dtype='float64', threads=torch.get_num_threads(),
                sample='the same batch, best of %d steps' % repeat)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=128)
    ap.add_argument('--steps', type=int, default=50)
    a = ap.parse_args()
    from eventclip_amd import _lib, clip as eclip, train
    from eventclip_amd.clip_cls import FSCLIPClassifier
    dev = _lib.require_gpu()
    cfg = eclip.arch_config('ViT-B/32', layers=1, text_layers=1, embed_dim=768)       # only the head is exercised
    model = eclip.CLIP(cfg, eclip.random_state_dict(cfg, seed=0)).cuda()
    # every device timing first, the CPU legs after: the oracle's torch-CPU worker threads keep spinning for a
    # while after their last parallel region and slow the launching thread of whatever is timed next
    cases = []
    for name, T, K in (('n_imagenet', 2, 1000), ('n_caltech', 10, 101), ('n_cars', 1, 2)):
        for kind in ('text-identity', 'text-trans'):
            cd = dict(clip_model=model, prompt='a point cloud image of a {}', class_names=[f'c{i}' for i in range(K)],
                      agg_func='mean', class_tokens=eclip.synthetic_tokens(K))
            ad = dict(adapter_type=kind, residual=0.95)
            if kind == 'text-trans':
                ad.update(in_dim=768, d_model=256, num_heads=4, ffn_dim=1024, norm_first=True, num_layers=2)
            clf = FSCLIPClassifier(adapter_dict=ad, clip_dict=cd, loss_dict=dict(use_logits_loss=True, use_probs_loss=False)).cuda()
            tr = (train.TextFeatTrainer if kind == 'text-identity' else train.AdapterTrainer)(clf, lr=2e-5, total_steps=10 ** 6)
            torch.manual_seed(0)
            B = a.batch
            feats = torch.randn(B, T, 768, device=dev)
            valid = torch.ones(B, T, dtype=torch.bool, device=dev)
            labels = torch.randint(0, K, (B,), device=dev)
            for _ in range(5):
                tr.step(feats, valid, labels)
            torch.cuda.synchronize()
            # two timed passes, the faster one counts (a one-off stall of the host -- an allocator refill, a code
            # object's first use -- would otherwise be most of a 50-step sum); the slowest single call is reported
            dt, worst = float('inf'), 0.0
            for _ in range(2):
                t0 = time.perf_counter()
                for _ in range(a.steps):
                    t1 = time.perf_counter()
                    loss = tr.step(feats, valid, labels)
                    worst = max(worst, time.perf_counter() - t1)
                torch.cuda.synchronize()
                dt = min(dt, (time.perf_counter() - t0) / a.steps)
            cases.append((kind, name, B, T, K, dt, float(loss), clf, feats, valid, labels, worst))
    for kind, name, B, T, K, dt, loss, clf, feats, valid, labels, worst in cases:
        cb = cpu_step(kind, clf, feats, valid, labels)
        cb['speedup'] = round(cb['ms_per_step'] / (dt * 1e3), 1)
        print(json.dumps(dict(adapter_type=kind, geometry=name, batch=B, views=T, classes=K,
                              ms_per_step=round(dt * 1e3, 3), samples_per_s=round(B / dt, 1),
                              loss=round(loss, 4), slowest_call_ms=round(worst * 1e3, 3), cpu_baseline=cb)), flush=True)


if __name__ == '__main__':
    main()
