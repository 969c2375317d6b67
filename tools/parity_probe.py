"""Logit error of the HIP path against the fp32 oracle, beside the reference's own fp16 GPU arithmetic (emulated
in the oracle), on weights whose features are input-dependent.  Prints, per case: the input-dependent share
of the feature norm, max|err| / max|logit|, and the CENTRED error max|err| / max|logit - mean over views|
for (a) the HIP path and (b) the fp16-reference emulation.

    python tools/parity_probe.py [--arch ViT-L/14] [--qk 6] [--branch 4]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--arch', default='ViT-L/14')
    ap.add_argument('--qk', type=float, nargs='+', default=[1.0, 2.5])
    ap.add_argument('--branch', type=float, nargs='+', default=[1.0, 4.0])
    ap.add_argument('--dtype', default='float16')
    ap.add_argument('--blob', type=float, default=0.7, help='share of the events in a Gaussian blob (structured frames)')
    ap.add_argument('--attn-variant', type=int, default=None, help='diagnostic build: 1 = the round-1/2 attention block')
    a = ap.parse_args()
    if a.attn_variant is not None:
        os.environ.setdefault('EVENTCLIP_HIP_LIB', os.path.join(ROOT, 'eventclip_amd', 'libeventclip_hip_diag.so'))
        import ctypes
        ctypes.CDLL(os.environ['EVENTCLIP_HIP_LIB']).ec_attn_set_variant(a.attn_variant)
    from eventclip_amd import clip as eclip
    from eventclip_amd.clip_cls import ZSCLIPClassifier
    from eventclip_amd.event2img import Event2ImagePipeline
    from eventclip_amd.synthetic import make_batch
    import test_configs_gpu as tc
    g, qa = tc.quantize_args('n_caltech', 10, grayscale=False)
    cfg = eclip.arch_config(a.arch, text_layers=2)
    tokens = eclip.synthetic_tokens(101, seed=5)
    evs = make_batch(3, [200000, 47000, 111000], g['resolution'], seed=5, blob_frac=a.blob)
    for qk, br in zip(a.qk, a.branch):
        sd = eclip.random_state_dict(cfg, seed=35, qk_gain=qk, branch_gain=br)
        m = eclip.CLIP(cfg, sd, dtype=a.dtype).cuda().eval()
        model = ZSCLIPClassifier(clip_dict=dict(clip_model=m, prompt='a point cloud image of a {}',
                                                class_names=[str(i) for i in range(101)],
                                                agg_func='mean', class_tokens=tokens)).cuda().eval()
        pipe = Event2ImagePipeline(g['resolution'], g['max_n'], qa, n_px=cfg['image_size'], patch=cfg['patch'], kpad=m.kpad)
        out = model(pipe(evs))
        want, feats = tc.oracle_forward(evs, g['resolution'], qa, cfg, sd, tokens, 10, 'mean')
        emu, _ = tc.oracle_forward(evs, g['resolution'], qa, cfg, sd, tokens, 10, 'mean', emulate='fp16_reference')
        share = float((feats - feats.mean(0)).norm() / feats.norm())
        line = f'{a.arch} qk {qk} branch {br} {a.dtype}: input-dependent share of the features {share:.3f};'
        for name, o in (('hip', {k: v.cpu() for k, v in out.items()}), ('fp16-reference emulation', emu)):
            e = tc.logit_errors(o, want)
            line += f' {name}: max-normalised {e["full_logits"][0]:.2e} / centred {e["full_logits"][1]:.2e}' \
                    f' (aggregated {e["logits"][0]:.2e} / {e["logits"][1]:.2e});'
        top1 = bool(torch.equal(out['logits'].argmax(-1).cpu(), want['logits'].argmax(-1)))
        top5 = bool(torch.equal(out['logits'].topk(5, -1).indices.cpu(), want['logits'].topk(5, -1).indices))
        print(line, f'top-1 agree {top1}, top-5 (ordered) agree {top5}', flush=True)


if __name__ == '__main__':
    main()
