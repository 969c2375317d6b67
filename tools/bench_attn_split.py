"""ec_attention_split at the bench shape (2560 sequences x 16 heads, S = 257; round 5): the hi + lo fp16 kernel on the 16-bit
matrix instruction against the fp32 kernel on v_mfma_f32_16x16x4_f32 (EC_ATTN_SPLIT_F32=1 selects the latter), and the
default 16-bit attention kernel beside them.

    python tools/bench_attn_split.py [--n-seq 2560] [--S 257];   EC_ATTN_SPLIT_F32=1 python tools/bench_attn_split.py
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# EC_ATTN_SPLIT_F32 is read by the DIAGNOSTIC build only since round 6 (the product library's kernel choice never depends on the environment)
os.environ.setdefault('EVENTCLIP_HIP_LIB', os.path.join(ROOT, 'eventclip_amd', 'libeventclip_hip_diag.so'))
from eventclip_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--n-seq', type=int, default=2560)
ap.add_argument('--S', type=int, default=257)
a = ap.parse_args()
lib = _lib.lib()
heads, W, S, n = 16, 1024, a.S, a.n_seq
torch.manual_seed(0)
qkv = torch.randn(n * S, 3 * W, device='cuda') * 1.5
pair = torch.empty((2, n * S, 3 * W), dtype=torch.float16, device='cuda')
pair[0] = qkv.half()
pair[1] = (qkv - pair[0].float()).half()
hi = torch.empty(n * S, W, dtype=torch.float16, device='cuda')
lo = torch.empty_like(hi)


def split():
    _lib.check(lib.ec_attention_split(_lib.ptr(pair[0]), _lib.ptr(pair[1]), _lib.ptr(hi), _lib.ptr(lo), n, S, W, heads, 0, _lib.EC_F16,
                                      _lib.stream_ptr()))


def plain():
    _lib.check(lib.ec_attention(_lib.ptr(pair[0]), _lib.ptr(hi), n, S, W, heads, 0, _lib.EC_F16, _lib.stream_ptr()))


for name, fn in (('ec_attention_split', split), ('ec_attention (16-bit)', plain)):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f'S={S} n_seq={n} {name:24s} {"(fp32 kernel)" if os.environ.get("EC_ATTN_SPLIT_F32") and "split" in name else "":14s}: {ms:.3f} ms = '
          f'{4.0 * S * S * 64 * heads * n / ms / 1e9:5.0f} TFLOP/s', flush=True)
split()
n_ref = 4
j = (pair[0][:n_ref * S].double() + pair[1][:n_ref * S].double())
q, k, v = j.view(n_ref, S, 3, heads, 64).permute(2, 0, 3, 1, 4)
want = (((q * 0.125) @ k.transpose(-1, -2)).softmax(-1) @ v).permute(0, 2, 1, 3).reshape(n_ref * S, W)
got = hi[:n_ref * S].double() + lo[:n_ref * S].double()
print(f'  error of hi + lo against float64: {float((got - want).abs().max() / want.abs().max()):.2e}')
