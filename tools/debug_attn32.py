"""Debug helper for the 32-query-tile attention kernel (diagnostic build, variant 5): structured inputs."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('EVENTCLIP_HIP_LIB', os.path.join(ROOT, 'eventclip_amd', 'libeventclip_hip_diag.so'))
from eventclip_amd import _lib  # noqa: E402

h = ctypes.CDLL(os.environ['EVENTCLIP_HIP_LIB'])
C = 0.125 * 1.4426950408889634


def run(qkv, S, heads, variant):
    W = heads * 64
    n_seq = qkv.shape[0] // S
    out = torch.full((n_seq * S, W), float('nan'), dtype=torch.float16, device='cuda')
    h.ec_attn_set_variant(variant)
    qin = qkv.clone()
    qin[:, :W] = (qkv[:, :W].float() * C).half()
    _lib.check(_lib.lib().ec_attention_scaled_q(_lib.ptr(qin), _lib.ptr(out), n_seq, S, W, heads, 0, S, _lib.EC_F16,
                                                _lib.stream_ptr()))
    torch.cuda.synchronize()
    h.ec_attn_set_variant(0)
    return out.float()


def ref(qkv, S, heads):
    W = heads * 64
    n_seq = qkv.shape[0] // S
    q, k, v = qkv.float().view(n_seq, S, 3, heads, 64).permute(2, 0, 3, 1, 4)
    att = (q * 0.125) @ k.transpose(-1, -2)
    return (att.softmax(-1) @ v).permute(0, 2, 1, 3).reshape(n_seq * S, W)


for S in (32, 64, 33, 50, 257):
    heads = 1
    W = 64
    torch.manual_seed(S)
    print(f'==== S={S}')
    # 1. q = 0: uniform attention, output = mean of V
    qkv = torch.zeros(S, 3 * W, device='cuda')
    qkv[:, 2 * W:] = torch.randn(S, W, device='cuda')
    qkv = qkv.half()
    got, want = run(qkv, S, heads, 5), ref(qkv, S, heads)
    e = (got - want).abs()
    print('uniform attention: max err', float(e.max()), 'rows bad', int((e.max(1).values > 1e-2).sum()), 'cols bad', int((e.max(0).values > 1e-2).sum()))
    if float(e.max()) > 1e-2:
        # V one-hot in dim: v[key][d] = (d == key % 64) -> mean = count/S per dim
        qkv = torch.zeros(S, 3 * W, device='cuda')
        qkv[:, 2 * W:] = torch.eye(64, device='cuda')[torch.arange(S) % 64] * torch.arange(1, S + 1, device='cuda')[:, None].float()
        qkv = qkv.half()
        got, want = run(qkv, S, heads, 5), ref(qkv, S, heads)
        print(' one-hot V row 0 got', (got[0] * S).round().tolist()[:40])
        print(' one-hot V row 0 want', (want[0] * S).round().tolist()[:40])
    # 2. random everything
    qkv = (torch.randn(S, 3 * W, device='cuda') * 1.0).half()
    got, want = run(qkv, S, heads, 5), ref(qkv, S, heads)
    e = (got - want).abs()
    print('random: max err', float(e.max()), 'rows bad', (e.max(1).values > 1e-2).nonzero().flatten().tolist()[:40])
    old = run(qkv, S, heads, 0)
    print('old kernel max err', float((old - want).abs().max()))
