"""A/B of the attention kernel's block variants in ONE process (diagnostic build), interleaved rounds.

    python -m eventclip_amd.build --diag && python tools/bench_attn.py [--S 257 577] [--rounds 5]

variant 0 = the product kernel (round 3: maximum subtracted by the MFMA's C operand, deferred rescale, row sum on
the matrix pipe), variant 1 = the round-1/2 block (per-block maximum, vector-ALU row sum).  Prints ms per launch
(median and min over the rounds), TFLOP/s, and the maximum difference of each variant from an fp32 torch
reference on the same data.
"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('EVENTCLIP_HIP_LIB', os.path.join(ROOT, 'eventclip_amd', 'libeventclip_hip_diag.so'))
from eventclip_amd import _lib  # noqa: E402


def ref_attention(qkv, n_seq, S, W, heads):
    q, k, v = qkv.float().view(n_seq, S, 3, heads, 64).permute(2, 0, 3, 1, 4)
    att = (q * 0.125) @ k.transpose(-1, -2)
    return (att.softmax(-1) @ v).permute(0, 2, 1, 3).reshape(n_seq * S, W)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--S', type=int, nargs='+', default=[257, 577])
    ap.add_argument('--n-seq', type=int, default=256)
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--scale', type=float, default=1.0, help='std of the synthetic q / k / v')
    ap.add_argument('--variants', type=int, nargs='+', default=[0, 1])
    ap.add_argument('--scaled-q', action='store_true',
                    help='the tower\'s entry point (ec_attention_scaled_q: q pre-multiplied by log2(e) / 8); variants 3 / 4 '
                         '(s_setprio around the MFMA groups / around the exponentials) only exist for it')
    a = ap.parse_args()
    h = ctypes.CDLL(os.environ['EVENTCLIP_HIP_LIB'])
    heads = 16
    W = heads * 64
    for S in a.S:
        torch.manual_seed(S)
        qkv = (torch.randn(a.n_seq * S, 3 * W, device='cuda') * a.scale).half()
        out = torch.empty(a.n_seq * S, W, dtype=torch.float16, device='cuda')
        flops = 4.0 * S * S * 64 * heads * a.n_seq
        # reference on a few sequences only (memory)
        n_ref = 8
        want = ref_attention(qkv[:n_ref * S], n_ref, S, W, heads)

        if a.scaled_q:      # the reference stays the plain softmax(q k^T / 8) v of the unscaled q
            qkv_in = qkv.clone()
            qkv_in[:, :W] = (qkv[:, :W].float() * (0.125 * 1.4426950408889634)).half()
        else:
            qkv_in = qkv

        def launch():
            if a.scaled_q:
                _lib.check(_lib.lib().ec_attention_scaled_q(_lib.ptr(qkv_in), _lib.ptr(out), a.n_seq, S, W, heads, 0, S,
                                                            _lib.EC_F16, _lib.stream_ptr()))
            else:
                _lib.check(_lib.lib().ec_attention(_lib.ptr(qkv), _lib.ptr(out), a.n_seq, S, W, heads, 0, _lib.EC_F16,
                                                   _lib.stream_ptr()))
        times = {v: [] for v in a.variants}
        err = {}
        for v in a.variants:
            h.ec_attn_set_variant(v)
            for _ in range(3):
                launch()
            torch.cuda.synchronize()
            err[v] = float((out[:n_ref * S].float() - want).abs().max())
        for _ in range(a.rounds):
            for v in a.variants:
                h.ec_attn_set_variant(v)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    launch()
                e1.record()
                torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) / a.iters)
        for v in a.variants:
            t = sorted(times[v])
            med, mn = t[len(t) // 2], t[0]
            print(f'S={S} variant {v}: median {med:.4f} ms  min {mn:.4f} ms  {flops / med / 1e9:.0f} TFLOP/s  '
                  f'max|err| {err[v]:.2e}', flush=True)
        h.ec_attn_set_variant(0)


if __name__ == '__main__':
    main()
