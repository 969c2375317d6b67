"""End-to-end logit error of the HIP path against the fp32 CPU oracle at FULL tower depth (GPU box).

    python tools/precision_probe.py [--arch ViT-L/14] [--frames 12] [--dtype float16] [--seeds 5 6 7]

Prints max |d| / max |ref| of the image features, of full_logits and of the aggregated logits
(the metric of tests/test_configs_gpu.py) for each seed, with the text features taken from the
(split-precision) text tower as the classifiers do.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--arch', default='ViT-L/14')
    ap.add_argument('--frames', type=int, default=12)
    ap.add_argument('--classes', type=int, default=101)
    ap.add_argument('--dtype', default='float16')
    ap.add_argument('--seeds', type=int, nargs='+', default=[5, 6, 7])
    ap.add_argument('--layers', type=int, default=None)
    ap.add_argument('--text-layers', type=int, default=2)
    a = ap.parse_args()
    from eventclip_amd import clip as eclip
    from eventclip_amd.synthetic import GEOMETRY, make_events
    from oracle import clip_ref
    from oracle import events as oe
    from oracle import preprocess as op
    g = GEOMETRY['n_caltech']
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    for seed in a.seeds:
        kw = dict(text_layers=a.text_layers)
        if a.layers:
            kw['layers'] = a.layers
        cfg = eclip.arch_config(a.arch, **kw)
        sd = eclip.random_state_dict(cfg, seed=seed)
        frames, i = [], 0
        while len(frames) < a.frames:
            ev = make_events(3 * g['N'], g['resolution'], seed=1000 * seed + i)
            frames.extend(list(oe.events2frames(ev, 'event_count', 'event_histogram', shape=g['resolution'],
                                                N=g['N'], grayscale=False, count_non_zero=False,
                                                background_mask=True)))
            i += 1
        imgs = torch.from_numpy(op.preprocess(np.stack(frames[:a.frames]), cfg['image_size']))
        tokens = eclip.synthetic_tokens(a.classes, seed=seed)
        t0 = time.time()
        ref_f = clip_ref.encode_image(sd, cfg, imgs)
        ref_t = torch.nn.functional.normalize(clip_ref.encode_text(sd, cfg, tokens), dim=-1)
        t_cpu = time.time() - t0
        m = eclip.CLIP(cfg, sd, dtype=a.dtype).cuda().eval()
        f = m.encode_image(imgs.cuda()).cpu()
        t = torch.nn.functional.normalize(m.encode_text(tokens.cuda()), dim=-1).cpu()
        lref, lgot = 100. * ref_f @ ref_t.T, 100. * f @ t.T
        # mean over groups of 3 views (the aggregated logits of agg_func='mean')
        n3 = (a.frames // 3) * 3
        aref, agot = lref[:n3].reshape(-1, 3, a.classes).mean(1), lgot[:n3].reshape(-1, 3, a.classes).mean(1)
        ef = float((f - ref_f).abs().max() / ref_f.abs().max())
        e2 = float((f - ref_f).norm() / ref_f.norm())
        et = float((t - ref_t).abs().max() / ref_t.abs().max())
        el = float((lgot - lref).abs().max() / lref.abs().max())
        ea = float((agot - aref).abs().max() / lref.abs().max())
        print(f'{a.arch} {a.dtype} seed {seed}: feats {ef:.2e} (l2 {e2:.2e})  text {et:.1e}  full_logits {el:.2e}  '
              f'logits {ea:.2e}   max|logit| {float(lref.abs().max()):.1f}  (oracle {t_cpu:.1f} s)', flush=True)


if __name__ == '__main__':
    main()
