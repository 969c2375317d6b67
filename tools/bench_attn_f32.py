"""A / B of the split-precision towers' fp32 attention (ec_attention_f32): the fp32-MFMA kernel the product runs
against the round-1 vector-ALU kernel (diagnostic build, ec_attn_set_variant(6)), interleaved in one process.

    python tools/bench_attn_f32.py [n_seq] [S] [heads]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('EVENTCLIP_HIP_LIB', os.path.join(ROOT, 'eventclip_amd', 'libeventclip_hip_diag.so'))
from eventclip_amd import _lib  # noqa: E402

n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 512
S = int(sys.argv[2]) if len(sys.argv) > 2 else 257
heads = int(sys.argv[3]) if len(sys.argv) > 3 else 16
W = heads * 64
lib = _lib.lib()
qkv = torch.randn(n_seq * S, 3 * W, device='cuda') * 1.5
outs = {}
times = {0: [], 6: []}
for v in (0, 6):
    hi = torch.empty(n_seq * S, W, dtype=torch.float16, device='cuda')
    lo = torch.empty_like(hi)
    outs[v] = (hi, lo)


def run(v):
    lib.ec_attn_set_variant(v)
    hi, lo = outs[v]
    _lib.check(lib.ec_attention_f32(_lib.ptr(qkv), _lib.ptr(hi), _lib.ptr(lo), n_seq, S, W, heads, 0, _lib.EC_F16,
                                    _lib.stream_ptr()), 'ec_attention_f32')


for v in (0, 6):
    run(v)
torch.cuda.synchronize()
a = outs[0][0].double() + outs[0][1].double()
b = outs[6][0].double() + outs[6][1].double()
print(f'max |fp32-MFMA - vector-ALU| / max = {float((a - b).abs().max() / b.abs().max()):.2e}')
for _ in range(5):
    for v in (0, 6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            run(v)
        e1.record()
        torch.cuda.synchronize()
        times[v].append(e0.elapsed_time(e1) / 3)
fl = 4.0 * S * S * 64 * heads * n_seq
for v, t in times.items():
    t = sorted(t)
    print(f'{"fp32 MFMA (product)" if v == 0 else "vector ALU (round 1)":22s}: median {t[2]:8.3f} ms per {n_seq} sequences '
          f'x {heads} heads, S = {S}: {fl / t[2] / 1e9:6.1f} TFLOP/s')
lib.ec_attn_set_variant(0)
