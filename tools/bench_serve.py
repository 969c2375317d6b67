"""Serving latency of the classifier forward for a few frames at a time: every GEMM launch in one pass (default) vs
the low-latency mode (`CLIP(..., low_latency=True)`: under-filled launches K-batched, ec_gemm_args.ws).

    python tools/bench_serve.py [--arch ViT-L/14] [--classes 101]

One JSON line per batch geometry: milliseconds per request, host-synchronised (until the logits are back), for B
samples x T views that arrive as patches; and the largest logit difference between the two modes."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--arch', default='ViT-L/14')
    ap.add_argument('--classes', type=int, default=101)
    ap.add_argument('--requests', type=int, default=50)
    a = ap.parse_args()
    from eventclip_amd import _lib, clip as eclip
    from eventclip_amd.clip_cls import ZSCLIPClassifier
    dev = _lib.require_gpu()
    K = a.classes
    clfs = {}
    for mode in (False, True):
        cfg = eclip.arch_config(a.arch)
        model = eclip.CLIP(cfg, eclip.random_state_dict(cfg, 0), low_latency=mode).cuda().eval()
        clfs[mode] = ZSCLIPClassifier(clip_dict=dict(clip_model=model, prompt='a point cloud image of a {}',
                                                     class_names=[f'c{i}' for i in range(K)], agg_func='sum',
                                                     class_tokens=eclip.synthetic_tokens(K))).cuda().eval()
    model = clfs[False].model
    G = (model.cfg['image_size'] // model.cfg['patch']) ** 2
    for B, T in ((1, 1), (1, 4), (1, 10), (4, 10), (16, 10)):
        n = B * T
        torch.manual_seed(n)
        patches = (torch.randn(n, G, model.kpad, device=dev) * 0.5).to(model.compute_dtype)
        data = {'patches': patches, 'row_idx': torch.arange(n, dtype=torch.int32, device=dev).view(B, T),
                'valid_mask': torch.ones(B, T, dtype=torch.bool, device=dev)}

        def latency(fn):
            for _ in range(5):
                fn(data)['logits'].sum().item()
            t0 = time.perf_counter()
            for _ in range(a.requests):
                fn(data)['logits'].sum().item()
            return (time.perf_counter() - t0) / a.requests * 1e3
        with torch.no_grad():
            base, fast = latency(clfs[False]), latency(clfs[True])
            l0, l1 = clfs[False](data)['logits'], clfs[True](data)['logits']
        print(json.dumps(dict(arch=a.arch, samples=B, views=T, frames=n, classes=K, single_pass_ms=round(base, 3),
                              low_latency_ms=round(fast, 3),
                              max_logit_diff_rel=float((l0 - l1).abs().max() / l0.abs().max()))), flush=True)


if __name__ == '__main__':
    main()
