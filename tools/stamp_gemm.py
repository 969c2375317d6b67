"""Diagnostic: per-segment s_memtime stamps of the 2-phase GEMM (variant 10)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the stamp / timeline variants only exist in the diagnostic build (python -m eventclip_amd.build --diag)
os.environ.setdefault('EVENTCLIP_HIP_LIB', os.path.join(ROOT, 'eventclip_amd', 'libeventclip_hip_diag.so'))
from eventclip_amd import ops
M, N, K = 65792, 1024, 4096
A = torch.randn(M, K, device='cuda').half(); W = (torch.randn(N, K, device='cuda') / 64).half()
out = torch.zeros(M, N, device='cuda')
dbg = torch.zeros(2 * 64 * 8 * 2, device='cuda', dtype=torch.float32)   # 2 waves x 64 tiles x 8 stamps (u64)
for _ in range(3):
    ops.gemm(A, W, None, 'store32', out=out, variant=10, diag=dbg)
torch.cuda.synchronize()
st = dbg.cpu().numpy().view(np.uint64).reshape(2, 64, 8).astype(np.int64)
names = ['L_A start', 'reads issued', 'dma+vmcnt', 'barrier1', 'mma done', 'L_B start(bar2)', 'L_B done', 'mma2 done']
for w in range(2):
    print('group', w)
    d = np.diff(st[w].reshape(-1))[: 63 * 8]
    d = d.reshape(63, 8)          # d[t][i] = stamp[i+1]-stamp[i], last = next tile's start - mma2 done
    print('  mean cycles per segment (tiles 8..55):')
    seg = ['A:ds_read issue', 'A:dma issue+vmcnt', 'A:lgkmcnt+barrier', 'A:32 MFMA', 'A:barrier2', 'B:reads+dma+vmcnt', 'B:lgkm+bar+32 MFMA', 'B:barrier2']
    for i in range(8):
        print(f'    {seg[i]:24s} {d[8:56, i].mean():8.1f}  (min {d[8:56, i].min()}, max {d[8:56, i].max()})')
    print('  K-tile period', (st[w, 56, 0] - st[w, 8, 0]) / 48.0)
