"""The key-half workgroup pair (round 5; diagnostic build only: measured, not kept -- profiles/r5_attention.md) against the
one-workgroup kernel (ec_attention_scaled_q) at the sequence lengths whose K and V fill a CU's LDS, interleaved in one process.

    python -m eventclip_amd.build --diag && python tools/bench_attn_pair.py [--S 577] [--n-seq 256]
    EC_PAIR_DEBUG=2 ...   no exchange at all (compute only: wrong output)      EC_PAIR_DEBUG=1 ...   agent-scope write-through stores

Prints ms per launch (median over the rounds), both kernels' error against an fp32 torch reference on the same data, and
whether a q_rows = 1 call of the pair is a bit-exact prefix of its full call.
"""
import argparse
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
    ap.add_argument('--S', type=int, nargs='+', default=[577])
    ap.add_argument('--n-seq', type=int, default=256)
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--iters', type=int, default=20)
    a = ap.parse_args()
    import ctypes
    lib = _lib.lib()
    diag = ctypes.CDLL(os.environ['EVENTCLIP_HIP_LIB'])
    diag.ec_attention_pair_workspace_bytes.restype = ctypes.c_size_t
    diag.ec_attention_pair_workspace_bytes.argtypes = [ctypes.c_int] * 3
    diag.ec_attention_pair.restype = ctypes.c_int
    diag.ec_attention_pair.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 7 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    heads, W = 16, 1024
    for S in a.S:
        torch.manual_seed(S)
        qkv = torch.randn(a.n_seq * S, 3 * W, device='cuda').half()
        scaled = qkv.clone()
        scaled[:, :W] = (qkv[:, :W].float() * (0.125 * 1.4426950408889634)).half()
        out_a = torch.empty(a.n_seq * S, W, dtype=torch.float16, device='cuda')
        out_b = torch.empty_like(out_a)
        ws = torch.empty(diag.ec_attention_pair_workspace_bytes(a.n_seq, S, heads), dtype=torch.uint8, device='cuda')

        def one():
            _lib.check(lib.ec_attention_scaled_q(_lib.ptr(scaled), _lib.ptr(out_a), a.n_seq, S, W, heads, 0, S, _lib.EC_F16, _lib.stream_ptr()))

        def pair(q_rows=S, out=out_b):
            _lib.check(diag.ec_attention_pair(_lib.ptr(scaled), _lib.ptr(out), a.n_seq, S, W, heads, q_rows, 1, _lib.EC_F16, _lib.ptr(ws),
                                             ws.numel(), _lib.stream_ptr()))
        for fn in (one, pair):
            for _ in range(3):
                fn()
        torch.cuda.synchronize()
        times = {'one workgroup': [], 'pair': []}
        for _ in range(a.rounds):
            for name, fn in (('one workgroup', one), ('pair', pair)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                times[name].append(e0.elapsed_time(e1) / a.iters)
        n_ref = 8
        want = ref_attention(qkv[:n_ref * S], n_ref, S, W, heads)
        flops = 4.0 * S * S * 64 * heads * a.n_seq
        for name, o in (('one workgroup', out_a), ('pair', out_b)):
            t = sorted(times[name])[len(times[name]) // 2]
            err = float((o[:n_ref * S].float() - want).abs().max())
            print(f'S={S} n_seq={a.n_seq} {name:14s}: {t:.3f} ms = {flops / t / 1e9:5.0f} TFLOP/s; max |err| vs fp32 {err:.2e}', flush=True)
        rows1 = torch.empty(a.n_seq, W, dtype=torch.float16, device='cuda')
        pair(1, rows1)
        torch.cuda.synchronize()
        print('  q_rows = 1 bit-exact prefix of the full call:', bool(torch.equal(rows1, out_b.view(a.n_seq, S, W)[:, 0])),
              '; pair run twice bit-identical:', end=' ')
        keep = out_b.clone()
        pair()
        torch.cuda.synchronize()
        print(bool(torch.equal(keep, out_b)))


if __name__ == '__main__':
    main()
