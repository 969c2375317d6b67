"""Time the few-shot training steps on cached encoder features at the reference's batch
shapes: `text-identity` (configs/fsclip/text_adapter: only text_feats trains) and `text-trans`
(configs/fsclip/joint_adapter: TransformerAdapter d_model 256, 2 layers, 4 heads + text_feats), train_batch_size 128,
views T, D = 768.

    python tools/bench_fs_train.py [--batch 128] [--steps 50]

One JSON line per (adapter type, geometry): ms per step and samples/s.  Synthetic features, seeded weights."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402


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
            t0 = time.perf_counter()
            for _ in range(a.steps):
                loss = tr.step(feats, valid, labels)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / a.steps
            print(json.dumps(dict(adapter_type=kind, geometry=name, batch=B, views=T, classes=K,
                                  ms_per_step=round(dt * 1e3, 3), samples_per_s=round(B / dt, 1),
                                  loss=round(float(loss), 4))), flush=True)


if __name__ == '__main__':
    main()
