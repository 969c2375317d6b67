"""Throughput of the per-GPU shard of BASELINE.json configs[2..4] (full depth, synthetic data, seeded random
weights; configs[1] is bench.py).  Run on the GPU box:  python tools/bench_configs.py [2 3 4]

These are parity-test cases (tests/test_configs_gpu.py), not bench lines; the numbers document that the
same kernels hold up on the other geometries (S = 577, K = 1000, the adapter, 480 x 640 events).
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eventclip_amd import _lib  # noqa: E402
from eventclip_amd import clip as eclip  # noqa: E402
from eventclip_amd.clip_cls import FSCLIPClassifier, ZSCLIPClassifier  # noqa: E402
from eventclip_amd.event2img import Event2ImagePipeline  # noqa: E402
from eventclip_amd.synthetic import GEOMETRY, make_events  # noqa: E402

CASES = {
    2: dict(name='N-Cars few-shot adapter, ViT-L/14, batch 512 x 1 view', geo='n_cars', arch='ViT-L/14', B=512, T=1,
            n_ev=12500, K=2, adapter=0.8, max_n=None),
    3: dict(name='N-ImageNet zero-shot, ViT-L/14@336px, 256 samples x 2 views per GPU (batch 2048 / 8)',
            geo='n_imagenet', arch='ViT-L/14@336px', B=256, T=2, n_ev=140000, K=1000, adapter=None, max_n=None),
    4: dict(name='N-ImageNet few-shot adapter, ViT-L/14, 512 samples x 5 views per GPU (batch 4096 / 8)',
            geo='n_imagenet', arch='ViT-L/14', B=512, T=5, n_ev=350000, K=1000, adapter=0.95, max_n=350000),
}


def run(idx, steps=3):
    c = CASES[idx]
    g = GEOMETRY[c['geo']]
    qa = dict(max_imgs=c['T'], N=g['N'], split_method='event_count', convert_method='event_histogram',
              grayscale=True, count_non_zero=g['count_non_zero'], background_mask=g['background_mask'])
    cfg = eclip.arch_config(c['arch'])
    m = eclip.CLIP(cfg, eclip.random_state_dict(cfg, seed=idx), chunk=2560).cuda().eval()
    clip_dict = dict(clip_model=m, prompt='a point cloud image of a {}', class_names=[str(i) for i in range(c['K'])],
                     agg_func='mean', class_tokens=eclip.synthetic_tokens(c['K'], seed=idx))
    if c['adapter'] is None:
        model = ZSCLIPClassifier(clip_dict=clip_dict)
    else:
        model = FSCLIPClassifier(adapter_dict=dict(adapter_type='text-trans', in_dim=cfg['embed_dim'], d_model=256,
                                                   num_heads=4, ffn_dim=1024, norm_first=True, num_layers=2,
                                                   residual=c['adapter']),
                                 clip_dict=clip_dict, loss_dict=dict(use_logits_loss=True, use_probs_loss=False))
    model = model.cuda().eval()
    model.get_text_feats()
    pipe = Event2ImagePipeline(g['resolution'], c['max_n'] or g['max_n'], qa, n_px=cfg['image_size'], patch=cfg['patch'],
                               kpad=m.kpad)
    pipe.strict = False
    uniq = [make_events(c['n_ev'], g['resolution'], seed=100 * idx + i) for i in range(4)]
    events = torch.from_numpy(np.concatenate([uniq[i % 4] for i in range(c['B'])])).cuda()
    n_events = [c['n_ev']] * c['B']

    def step():
        return model(pipe(events, n_events))

    out = step()
    frames = int(out['valid_masks'].sum())
    torch.cuda.synchronize()
    _lib.profile_begin()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    prof = _lib.profile_end()
    top = sorted(prof, key=lambda e: -e['total_ms'])[:6]
    print(f'configs[{idx}] {c["name"]}: {frames} frames in {dt * 1e3:.1f} ms = {frames / dt:.0f} frames/s per GPU')
    print('   ' + ', '.join(f'{e["name"]} {e["total_ms"] / steps:.1f} ms' for e in top), flush=True)


if __name__ == '__main__':
    for i in ([int(a) for a in sys.argv[1:]] or [2, 3, 4]):
        run(i)
        torch.cuda.empty_cache()
