"""Where does the 16-bit path's logit error come from?  CPU experiment (torch fp32).

Runs the vision tower of oracle/clip_ref.py on a few event frames with the HIP path's 16-bit
rounding points switched on one group at a time (everything else fp32), and prints the relative
error of the zero-shot logits (max |d| / max |ref|, the metric of tests/test_configs_gpu.py) each
group causes alone, all of them together, and all but one.  The groups:

  patch   pixel values and conv1.weight rounded (the patch-embedding GEMM operands)
  h       ln_1 / ln_2 outputs rounded (A operands of QKV and c_fc)
  wqkv wout wfc1 wfc2   the four block weights rounded
  qkv     QKV GEMM output rounded (attention operands); in plans also 'qk' (q and k only) and 'v'
  p       softmax probabilities rounded (P operand of P.V)
  att     attention output rounded (A operand of out_proj)
  gelu    QuickGELU output rounded (A operand of c_proj)
  post    ln_post(CLS) output and visual.proj rounded

    python tools/rounding_budget.py [--frames 4] [--layers 24] [--dtype float16]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

GROUPS = ['patch', 'h', 'wqkv', 'wout', 'wfc1', 'wfc2', 'qkv', 'p', 'att', 'gelu', 'post']


def make_rounder(dtype):
    def r(x):
        return x.to(dtype).float()
    return r


def tower(sd, cfg, image, on, rnd, first_layer=0, last_layer=None):
    """encode_image with rounding applied for the groups in `on`, in blocks
    first_layer <= l < last_layer only (patch / post are not layer-gated)."""
    W, P, L = cfg['width'], cfg['patch'], cfg['layers']
    last_layer = L if last_layer is None else last_layer
    heads = W // 64

    def q(name, x, l=None):
        if callable(on):      # a plan: on(group, block) -> bool (block None for patch / post)
            return rnd(x) if on(name, l) else x
        if name not in on:
            return x
        if l is not None and not (first_layer <= l < last_layer):
            return x
        return rnd(x)

    x = F.conv2d(q('patch', image), q('patch', sd['visual.conv1.weight']), stride=P)
    x = x.reshape(x.shape[0], W, -1).permute(0, 2, 1)
    cls = sd['visual.class_embedding'].expand(x.shape[0], 1, W)
    x = torch.cat([cls, x], dim=1) + sd['visual.positional_embedding']
    x = F.layer_norm(x, (W,), sd['visual.ln_pre.weight'], sd['visual.ln_pre.bias'], 1e-5)
    N, S, _ = x.shape
    for l in range(L):
        p = f'visual.transformer.resblocks.{l}.'
        h = q('h', F.layer_norm(x, (W,), sd[p + 'ln_1.weight'], sd[p + 'ln_1.bias'], 1e-5), l)
        qkv = q('qkv', F.linear(h, q('wqkv', sd[p + 'attn.in_proj_weight'], l),
                                sd[p + 'attn.in_proj_bias']), l)
        qq, kk, vv = qkv.split(W, dim=-1)
        qq, kk, vv = q('qk', qq, l), q('qk', kk, l), q('v', vv, l)      # (finer than 'qkv': plans only)
        qq = qq.view(N, S, heads, 64).transpose(1, 2)
        kk = kk.view(N, S, heads, 64).transpose(1, 2)
        vv = vv.view(N, S, heads, 64).transpose(1, 2)
        att = (qq @ kk.transpose(-1, -2)) * 0.125
        # the kernel keeps exp(s - max) in 16 bits and divides by the fp32 row sum afterwards
        e = torch.exp(att - att.amax(dim=-1, keepdim=True))
        o = (q('p', e, l) @ vv) / e.sum(dim=-1, keepdim=True)
        o = q('att', o.transpose(1, 2).reshape(N, S, W), l)
        x = x + F.linear(o, q('wout', sd[p + 'attn.out_proj.weight'], l), sd[p + 'attn.out_proj.bias'])
        h = q('h', F.layer_norm(x, (W,), sd[p + 'ln_2.weight'], sd[p + 'ln_2.bias'], 1e-5), l)
        m = F.linear(h, q('wfc1', sd[p + 'mlp.c_fc.weight'], l), sd[p + 'mlp.c_fc.bias'])
        m = q('gelu', m * torch.sigmoid(1.702 * m), l)
        x = x + F.linear(m, q('wfc2', sd[p + 'mlp.c_proj.weight'], l), sd[p + 'mlp.c_proj.bias'])
    c = F.layer_norm(x[:, 0, :], (W,), sd['visual.ln_post.weight'], sd['visual.ln_post.bias'], 1e-5)
    return q('post', c) @ q('post', sd['visual.proj'])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=4)
    ap.add_argument('--layers', type=int, default=24)
    ap.add_argument('--arch', default='ViT-L/14')
    ap.add_argument('--dtype', default='float16')
    ap.add_argument('--seed', type=int, default=5)
    ap.add_argument('--classes', type=int, default=101)
    ap.add_argument('--qk', type=float, default=1.0, help='random_state_dict qk_gain (2.5: the parity tests)')
    ap.add_argument('--branch', type=float, default=1.0, help='random_state_dict branch_gain (4: the parity tests)')
    ap.add_argument('--blob', type=float, default=0.1, help='share of events in the Gaussian blob (0.7: the parity tests)')
    ap.add_argument('--mixes', default='', help='extra comma-separated mixes to evaluate, groups '
                    'joined by +, an optional @a:b restricts the rounding to blocks a..b-1; '
                    'each mix lists the groups that STAY rounded')
    ap.add_argument('--f16-weights', action='store_true',
                    help='every weight tensor rounded to the 16-bit type FIRST, reference included: weights that are '
                         'representable in 16 bit, as the parameters clip.load() returns on a GPU are')
    ap.add_argument('--mixes-only', action='store_true')
    a = ap.parse_args()
    from eventclip_amd import clip as eclip
    from eventclip_amd.synthetic import GEOMETRY, make_events
    from oracle import events as oe
    from oracle import preprocess as op
    torch.manual_seed(0)
    cfg = eclip.arch_config(a.arch, layers=a.layers)
    sd = {k: v.float() for k, v in eclip.random_state_dict(cfg, seed=a.seed, qk_gain=a.qk, branch_gain=a.branch).items()}
    if a.f16_weights:
        sd = {k: (v.to(getattr(torch, a.dtype)).float() if v.dim() >= 2 else v) for k, v in sd.items()}
    g = GEOMETRY['n_caltech']
    frames = []
    i = 0
    while len(frames) < a.frames:
        ev = make_events(2 * g['N'], g['resolution'], seed=300 + i, blob_frac=a.blob)
        f = oe.events2frames(ev, 'event_count', 'event_histogram', shape=g['resolution'], N=g['N'],
                             grayscale=False, count_non_zero=False, background_mask=True)
        frames.extend(list(f))
        i += 1
    imgs = torch.from_numpy(op.preprocess(np.stack(frames[:a.frames]), cfg['image_size']))
    gen = torch.Generator().manual_seed(11)
    text = F.normalize(torch.randn(a.classes, cfg['embed_dim'], generator=gen), dim=-1)
    rnd = make_rounder(getattr(torch, a.dtype))

    def logits(feats):
        return 100.0 * feats @ text.T

    with torch.no_grad():
        t0 = time.time()
        ref = tower(sd, cfg, imgs, set(), rnd)
        lref = logits(ref)
        print(f'reference: {time.time() - t0:.1f} s, |f| rms {float(ref.pow(2).mean().sqrt()):.3f} '
              f'max {float(ref.abs().max()):.3f}, max |logit| {float(lref.abs().max()):.2f}')

        def report(name, on, **kw):
            f = tower(sd, cfg, imgs, set(on), rnd, **kw)
            ef = float((f - ref).abs().max() / ref.abs().max())
            el = float((logits(f) - lref).abs().max() / lref.abs().max())
            e2 = float((f - ref).norm() / ref.norm())
            print(f'{name:34s} feats {ef:.2e}  l2 {e2:.2e}  logits {el:.2e}', flush=True)
            return el

        report('all rounded (the fast path)', GROUPS)
        for grp in ([] if a.mixes_only else GROUPS):
            report(f'only {grp}', [grp])
        for grp in ([] if a.mixes_only else GROUPS):
            report(f'all but {grp}', [x for x in GROUPS if x != grp])
        for mix in [m for m in a.mixes.split(',') if m]:
            if ';' in mix:
                # a plan: 'groups@a:b;groups@c:d;...' -- the union of (group, block range) pairs that stay rounded
                parts = []
                for part in mix.split(';'):
                    spec, _, rng = part.partition('@')
                    lo, hi = (rng.split(':') if rng else (0, a.layers))
                    parts.append((set(x for x in spec.split('+') if x), int(lo), int(hi)))

                def plan(name, l, parts=parts):
                    return any(name in g and (l is None or lo <= l < hi) for g, lo, hi in parts)
                f = tower(sd, cfg, imgs, plan, rnd)
                ef = float((f - ref).abs().max() / ref.abs().max())
                el = float((logits(f) - lref).abs().max() / lref.abs().max())
                print(f'plan {mix:60s} feats {ef:.2e}  logits {el:.2e}', flush=True)
                continue
            spec, _, rng = mix.partition('@')
            kw = {}
            if rng:
                lo, hi = rng.split(':')
                kw = dict(first_layer=int(lo), last_layer=int(hi))
            report(f'mix {mix}', [x for x in spec.split('+') if x], **kw)


if __name__ == '__main__':
    main()
