"""Attention kernel alone (ViT-L/14 shape), target for rocprofv3 --pmc runs.

    python3 tools/pmc_attn.py [S] [variant]     # variant needs the diagnostic build (1 = the round-1/2 block)
"""
import ctypes
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 2:
    os.environ.setdefault('EVENTCLIP_HIP_LIB', os.path.join(ROOT, 'eventclip_amd', 'libeventclip_hip_diag.so'))
from eventclip_amd import _lib  # noqa: E402
if len(sys.argv) > 2:
    ctypes.CDLL(os.environ['EVENTCLIP_HIP_LIB']).ec_attn_set_variant(int(sys.argv[2]))
n_seq, S, heads = 256, int(sys.argv[1]) if len(sys.argv) > 1 else 257, 16
W = heads * 64
qkv = (torch.randn(n_seq * S, 3 * W, device='cuda')).half()
out = torch.empty(n_seq * S, W, dtype=torch.float16, device='cuda')
for _ in range(5):
    _lib.check(_lib.lib().ec_attention(_lib.ptr(qkv), _lib.ptr(out), n_seq, S, W, heads, 0, _lib.EC_F16, _lib.stream_ptr()))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    _lib.check(_lib.lib().ec_attention(_lib.ptr(qkv), _lib.ptr(out), n_seq, S, W, heads, 0, _lib.EC_F16, _lib.stream_ptr()))
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f'attention S={S}: {ms:.3f} ms, {4.0*S*S*64*heads*n_seq/ms/1e9:.1f} TFLOP/s')
