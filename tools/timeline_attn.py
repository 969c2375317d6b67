"""Diagnostic: where an attention workgroup spends its time (s_memtime stamps of the diagnostic build:
start -> K / V staged -> done), and how the workgroups of one CU follow each other.

    python -m eventclip_amd.build --diag && python tools/timeline_attn.py [S] [n_seq]
"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('EVENTCLIP_HIP_LIB', os.path.join(ROOT, 'eventclip_amd', 'libeventclip_hip_diag.so'))
from eventclip_amd import _lib  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 257
n_seq = int(sys.argv[2]) if len(sys.argv) > 2 else 256
heads = 16
W = heads * 64
if len(sys.argv) > 3:   # block variant of the diagnostic build (1 = round-1/2 block)
    ctypes.CDLL(os.environ['EVENTCLIP_HIP_LIB']).ec_attn_set_variant(int(sys.argv[3]))
qkv = torch.randn(n_seq * S, 3 * W, device='cuda').half()
out = torch.empty(n_seq * S, W, dtype=torch.float16, device='cuda')
scaled = os.environ.get('ATTN_SCALED', '0') != '0'     # the tower's entry point (variant 5 = 32-query tiles needs it)
for _ in range(3):
    if scaled:
        _lib.check(_lib.lib().ec_attention_scaled_q(_lib.ptr(qkv), _lib.ptr(out), n_seq, S, W, heads, 0, S, _lib.EC_F16,
                                                    _lib.stream_ptr()))
    else:
        _lib.check(_lib.lib().ec_attention(_lib.ptr(qkv), _lib.ptr(out), n_seq, S, W, heads, 0, _lib.EC_F16, _lib.stream_ptr()))
torch.cuda.synchronize()
n = min(n_seq * heads, 65536)
host = np.zeros(n * 4, dtype=np.uint64)
h = ctypes.CDLL(os.environ['EVENTCLIP_HIP_LIB'])
assert h.ec_attn_stamps_read(host.ctypes.data_as(ctypes.c_void_p), n * 4) == 0
r = host.reshape(n, 4).astype(np.int64)
stage, comp = r[:, 1] - r[:, 0], r[:, 2] - r[:, 1]
print(f'S={S}: {n} workgroups; cycles (s_memtime = 100 MHz ticks x ... shader clock units)')
for name, v in (('staging (start -> barrier)', stage), ('compute (barrier -> done)', comp), ('total', stage + comp)):
    print(f'  {name:30s} mean {v.mean():9.0f}  p10 {np.percentile(v, 10):9.0f}  p50 {np.percentile(v, 50):9.0f}  p90 {np.percentile(v, 90):9.0f}')
span = r[:, 2].max() - r[:, 0].min()
print(f'  kernel span {span} ticks; sum of workgroup times / (span x 512 resident) = {float((stage + comp).sum()) / (span * 512):.2f}')
cu = ((r[:, 3] >> 32) & 15) << 16 | (r[:, 3] & 0xff00)
for c in np.unique(cu)[:2]:
    sel = np.argsort(r[cu == c, 0])
    rows = r[cu == c][sel]
    print('  CU', hex(int(c)), 'workgroups (start, staged, done) relative:', [(int(a - rows[0, 0]), int(b - rows[0, 0]), int(d - rows[0, 0])) for a, b, d, _ in rows[:8]])
