"""CPU model (torch fp32) of the image tower's TOLERANCE MODE with its rounding points, and of the same mode with the
lo products on gfx950's FP8 matrix path -- run BEFORE any kernel is written: the stopping rule
is that the worst of the eight (weight seed, event seed) draws of every BASELINE config must stay inside north_star's
1e-3 with the FP8 lo products, else nothing is built.

    python tools/tolerance_model.py --configs 0,1,2,3,4 --seeds 8 --variants f16,e4m3 > profiles/r6_fp8_model.txt

The model follows csrc/towers.hip run_blocks_folded: blocks l < B are split-operand blocks (LayerNorm of the fp32 stream
into hi + lo fp16 parts, every GEMM = x_hi.W16 + x_lo.W16 + x_hi.W_lo with fp32 accumulation; in the first A of them q | k | v,
the attention output and QuickGELU's output are carried as hi + lo and attention is exact, in the others they are rounded to
fp16 and attention runs with fp16 probabilities), blocks l >= B are the default chain (hi plane of the stream as the GEMM
operand, LayerNorm finished behind the product from statistics of the hi plane, gain and softmax scale folded into the
weights before their rounding, fp16 q | k | v / probabilities / attention output / MLP activation).  Patch embedding,
ln_post @ proj and the logits are exact (hi + lo operands on the GPU).

Variants of the lo products (x_lo . W and x . W_lo):
  f16    as shipped: both operands fp16
  e4m3   x_lo (and W_lo) quantised to FP8 e4m3 at ONE power-of-two scale per tensor (2^12 for lo parts of activations:
         saturates beyond |x| = 256; per matrix for weights), multiplied by the e4m3 copy of the other operand (W16 / x_hi
         quantised the same way): v_mfma_f32_16x16x128_f8f6f4 at twice the f16 rate, a uniform scale operand
  mx8    the same with an e8m0 scale per 32 K-elements (the block-scaled form of the instruction)
  mx6    e2m3 (FP6) with an e8m0 scale per 32 K-elements: four times the f16 rate
--lo8 chooses which lo products take the variant: h (x_lo of QKV / c_fc), att (attention output into out_proj), gelu (MLP
activation into c_proj), w (the x_hi . W_lo products).  Everything is compared with the fp32 oracle logits shipped in
tests/golden/configs_oracle_*_signal.npz through the same logit_errors as tests/test_configs_gpu.py.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

ATTN_Q_SCALE = 0.125 * 1.4426950408889634
EXACT = {'qk', 'v', 'p', 'att', 'gelu'}
EXACT_REST = set()


def h16(x):
    return x.half().float()


def split16(x):
    hi = h16(x)
    return hi, h16(x - hi)


def q_e4m3(x, scale_log2):
    """e4m3 (fn: max 448, saturating) at one power-of-two scale; returns the dequantised values"""
    s = 2.0 ** scale_log2
    return (x * s).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float() / s


def _block_scale(x, emax):
    """e8m0 scale per 32 consecutive K elements (last dim): 2^(floor(log2 max|x|) - emax)"""
    K = x.shape[-1]
    b = x.reshape(*x.shape[:-1], K // 32, 32)
    m = b.abs().amax(dim=-1, keepdim=True).clamp_min(2.0 ** -126)
    return b, torch.exp2(torch.floor(torch.log2(m)) - emax)


def q_mx8(x):
    b, s = _block_scale(x, 8)
    return ((b / s).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float() * s).reshape(x.shape)


def q_mx6(x):
    """e2m3: 1 sign, 2 exponent (bias 1), 3 mantissa bits: normals 1 .. 7.5, subnormal step 0.125"""
    b, s = _block_scale(x, 2)
    v = (b / s).clamp(-7.5, 7.5)
    e = torch.floor(torch.log2(v.abs().clamp_min(1e-30))).clamp(0, 2)
    step = torch.exp2(e - 3)
    return (torch.round(v / step) * step * s).reshape(x.shape)


class Variant:
    def __init__(self, kind, lo8):
        self.kind, self.lo8 = kind, set(lo8)

    def quant_act_lo(self, x):          # lo part of an activation (magnitude ~ 2^-11 of the activation)
        return {'e4m3': lambda t: q_e4m3(t, 12), 'mx8': q_mx8, 'mx6': q_mx6}[self.kind](x)

    def quant_full(self, x, is_weight):  # the e4m3 / e2m3 copy of a full-size operand (x_hi or W16)
        if self.kind == 'e4m3':
            m = float(x.abs().max())
            sl = 8 - int(np.ceil(np.log2(max(m, 1e-30)))) if is_weight else 0    # weights: max into [128, 256); activations: 2^0
            return q_e4m3(x, sl)
        return {'mx8': q_mx8, 'mx6': q_mx6}[self.kind](x)

    def lo_product(self, which, a_lo, w16):
        """a_lo . w16^T for the activation lo part `which` in ('h', 'att', 'gelu')"""
        if self.kind == 'f16' or which not in self.lo8:
            return F.linear(a_lo, w16)
        return F.linear(self.quant_act_lo(a_lo), self.quant_full(w16, True))

    def wlo_product(self, a_hi, w_lo):
        if w_lo is None:
            return 0.0
        if self.kind == 'f16' or 'w' not in self.lo8:
            return F.linear(a_hi, w_lo)
        m = float(w_lo.abs().max())
        sl = 8 - int(np.ceil(np.log2(max(m, 1e-30))))
        wq = q_e4m3(w_lo, sl) if self.kind == 'e4m3' else {'mx8': q_mx8, 'mx6': q_mx6}[self.kind](w_lo)
        return F.linear(self.quant_full(a_hi, False), wq)


def attention16(q, k, v, heads, scale):
    """the 16-bit kernel: fp16 q, k, v; fp32 scores; probabilities rounded to fp16 (P operand), their sum from the rounded
    values; output in fp32 (the caller rounds)"""
    N, S, W = q.shape
    q = q.view(N, S, heads, 64).transpose(1, 2)
    k = k.view(N, S, heads, 64).transpose(1, 2)
    v = v.view(N, S, heads, 64).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) * scale
    p = h16(torch.exp2(s - s.amax(dim=-1, keepdim=True)))
    o = (p @ v) / p.sum(dim=-1, keepdim=True)
    return o.transpose(1, 2).reshape(N, S, W)


def attention_exact(q, k, v, heads):
    N, S, W = q.shape
    q = q.view(N, S, heads, 64).transpose(1, 2) * 0.125
    k = k.view(N, S, heads, 64).transpose(1, 2)
    v = v.view(N, S, heads, 64).transpose(1, 2)
    return ((q @ k.transpose(-1, -2)).softmax(dim=-1) @ v).transpose(1, 2).reshape(N, S, W)


def parse_set(spec, L):
    """'0-7,22-23' -> {0..7, 22, 23}; negative numbers count from the end ('-2--1' = the last two blocks)"""
    out = set()
    for part in [p for p in spec.split(',') if p]:
        m = __import__('re').fullmatch(r'(-?\d+)(?:-(-?\d+))?', part)
        lo = int(m.group(1))
        hi = int(m.group(2)) if m.group(2) is not None else lo
        lo, hi = (lo + L if lo < 0 else lo), (hi + L if hi < 0 else hi)
        out |= set(range(lo, hi + 1))
    return out


@torch.no_grad()
def tower(sd, cfg, image, B, A, var, exact16=False):
    """encode_image in the tolerance mode (B split-operand blocks, A of them with exact attention) under `var`.
    B / A may be SETS of block indices (attribution experiments: which blocks' roundings matter)."""
    W, P, L = cfg['width'], cfg['patch'], cfg['layers']
    heads = W // 64
    x = F.conv2d(image, sd['visual.conv1.weight'], stride=P)
    x = x.reshape(x.shape[0], W, -1).permute(0, 2, 1)
    x = torch.cat([sd['visual.class_embedding'].expand(x.shape[0], 1, W), x], dim=1) + sd['visual.positional_embedding']
    x = F.layer_norm(x, (W,), sd['visual.ln_pre.weight'], sd['visual.ln_pre.bias'], 1e-5)

    def wparts(w):
        w16 = h16(w)
        return w16, (None if exact16 else h16(w - w16))

    def gemm_split(a_hi, a_lo, which, w, bias):
        w16, w_lo = wparts(w)
        y = F.linear(a_hi, w16, bias)
        if a_lo is not None:
            y = y + var.lo_product(which, a_lo, w16)
        return y + var.wlo_product(a_hi, w_lo)

    for l in range(L):
        p = f'visual.transformer.resblocks.{l}.'
        g1, b1, g2, b2 = sd[p + 'ln_1.weight'], sd[p + 'ln_1.bias'], sd[p + 'ln_2.weight'], sd[p + 'ln_2.bias']
        wqkv, bqkv = sd[p + 'attn.in_proj_weight'], sd[p + 'attn.in_proj_bias']
        wout, bout = sd[p + 'attn.out_proj.weight'], sd[p + 'attn.out_proj.bias']
        wfc1, bfc1 = sd[p + 'mlp.c_fc.weight'], sd[p + 'mlp.c_fc.bias']
        wfc2, bfc2 = sd[p + 'mlp.c_proj.weight'], sd[p + 'mlp.c_proj.bias']
        if (l in B) if isinstance(B, (set, frozenset)) else l < B:
            pa = (l in A) if isinstance(A, (set, frozenset)) else l < A
            h_hi, h_lo = split16(F.layer_norm(x, (W,), g1, b1, 1e-5))
            qkv = gemm_split(h_hi, h_lo, 'h', wqkv, bqkv)
            # EXACT: which of the pa blocks' activations are carried as hi + lo / computed exactly (attribution experiments;
            # the shipped mode: all of them).  qk / v: q, k / v as hi + lo; p: exact softmax probabilities; att / gelu: the
            # attention output / the MLP activation as hi + lo into the next GEMM
            ex = EXACT if pa else EXACT_REST
            qq, kk, vv = qkv.split(W, dim=-1)
            r2 = (lambda t: sum(split16(t)))
            qq, kk = (r2(qq), r2(kk)) if 'qk' in ex else (h16(qq), h16(kk))
            vv = r2(vv) if 'v' in ex else h16(vv)
            att = attention_exact(qq, kk, vv, heads) if 'p' in ex else attention16(qq * 0.125, kk, vv, heads, 1.4426950408889634)
            a_hi, a_lo = split16(att) if 'att' in ex else (h16(att), None)
            x = x + gemm_split(a_hi, a_lo, 'att', wout, bout)
            h_hi, h_lo = split16(F.layer_norm(x, (W,), g2, b2, 1e-5))
            m = gemm_split(h_hi, h_lo, 'h', wfc1, bfc1)
            m = m * torch.sigmoid(1.702 * m)
            m_hi, m_lo = split16(m)
            pa = 'gelu' in ex
            x = x + gemm_split(m_hi, m_lo if pa else None, 'gelu', wfc2, bfc2)
            continue
        # ---- default chain: raw hi plane as the operand, LayerNorm finished behind the product ----
        def folded(xr, gamma, beta, w, bias, qrows=0):
            x_hi = h16(xr)
            mean = x_hi.mean(-1, keepdim=True)
            rstd = torch.rsqrt((x_hi * x_hi).mean(-1, keepdim=True) - mean * mean + 1e-5)
            wf = w * gamma
            bf = bias + w @ beta
            if qrows:
                wf = torch.cat([wf[:qrows] * ATTN_Q_SCALE, wf[qrows:]])
                bf = torch.cat([bf[:qrows] * ATTN_Q_SCALE, bf[qrows:]])
            w16 = h16(wf)
            return rstd * F.linear(x_hi, w16) - rstd * mean * w16.sum(-1) + bf
        qq, kk, vv = h16(folded(x, g1, b1, wqkv, bqkv, qrows=W)).split(W, dim=-1)
        att = h16(attention16(qq, kk, vv, heads, 1.0))
        x = x + F.linear(att, h16(wout), bout)
        m = folded(x, g2, b2, wfc1, bfc1)
        m = h16(m * torch.sigmoid(1.702 * m))
        x = x + F.linear(m, h16(wfc2), bfc2)
    c = F.layer_norm(x[:, 0, :], (W,), sd['visual.ln_post.weight'], sd['visual.ln_post.bias'], 1e-5)
    return c @ sd['visual.proj']


def main():
    import config_cases as cc
    import test_configs_gpu as tc
    from eventclip_amd import clip as eclip
    from oracle import clip_ref
    ap = argparse.ArgumentParser()
    ap.add_argument('--configs', default='0,1,2,3,4')
    ap.add_argument('--seeds', type=int, default=8)
    ap.add_argument('--draws', default=None)
    ap.add_argument('--variants', default='f16,e4m3')
    ap.add_argument('--lo8', default='h,att,gelu,w')
    ap.add_argument('--weights', default='signal')
    ap.add_argument('--exact-rest', default='', help='the same for the split-operand blocks behind them (shipped: none)')
    ap.add_argument('--exact', default='qk,v,p,att,gelu', help='activations carried exactly in the fp32-attention blocks (attribution)')
    ap.add_argument('--blocks', default=None, help="B:A instead of eventclip_amd.clip.TOLERANCE_MODE; or block SETS '0-7,22-23:0-4'")
    a = ap.parse_args()
    global EXACT, EXACT_REST
    EXACT = set(x for x in a.exact.split(',') if x)
    EXACT_REST = set(x for x in a.exact_rest.split(',') if x)
    variants = [Variant(k, a.lo8.split(',')) for k in a.variants.split(',')]
    rows = {}
    real = clip_ref.encode_image
    t0 = time.time()
    for c in [int(x) for x in a.configs.split(',')]:
        draws = [int(d) for d in a.draws.split(',')] if a.draws else range(a.seeds)
        for d in draws:
            inp = cc.build_inputs(c, a.weights, d)
            gold = cc.load_golden(c, a.weights, d)
            want = dict(full_logits=torch.from_numpy(gold['full_logits']), logits=torch.from_numpy(gold['logits']),
                        valid_masks=torch.from_numpy(gold['valid_masks']))
            if a.blocks and ('-' in a.blocks or ',' in a.blocks):
                Bs, As = a.blocks.split(':')
                B, A = parse_set(Bs, inp['cfg']['layers']), parse_set(As, inp['cfg']['layers'])
            elif a.blocks:
                B, A = (int(v) for v in a.blocks.split(':'))
            else:
                kw = eclip.tolerance_mode_kwargs(inp['cfg'])
                B, A = kw['image_precise_blocks'], kw['image_precise_attn_blocks']
            for var in variants:
                def enc(sd, cfg, imgs, emulate=None, var=var):
                    return tower({k: v.float() for k, v in sd.items()}, cfg, imgs.float(), B, A, var,
                                 exact16=(a.weights == 'signal16'))
                clip_ref.encode_image = enc
                try:
                    out, _ = cc.oracle_case(inp)
                finally:
                    clip_ref.encode_image = real
                e = tc.logit_errors(out, want)
                rows.setdefault((var.kind, c), []).append(e['full_logits'][0])
                print(f'[model {var.kind} {a.blocks or str(B) + ":" + str(A)}] configs[{c}] draw {d}: full_logits {e["full_logits"][0]:.2e} '
                      f'(centred {e["full_logits"][1]:.2e}), logits {e["logits"][0]:.2e}  ({time.time() - t0:.0f} s)', flush=True)
    print()
    print(f'CPU model of the tolerance mode, lo products as f16 / FP8 ({a.lo8}), weights = {a.weights}: full_logits max |err| / max |logit|')
    print('variant | config | median | worst | draws inside 1e-3')
    for (k, c), v in rows.items():
        v = np.asarray(v)
        print(f'{k} | configs[{c}] | {np.median(v):.2e} | {v.max():.2e} (draw {int(v.argmax())}) | {int((v < 1e-3).sum())} / {len(v)}')


if __name__ == '__main__':
    main()
