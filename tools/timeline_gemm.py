"""Diagnostic: per-CU timeline of the default GEMM (variant 16): how long a CU spends between one
workgroup's last store and the next one's first MFMA.  Run on the GPU box.

    python tools/timeline_gemm.py [N] [K] [epilogue]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the stamp / timeline variants only exist in the diagnostic build (python -m eventclip_amd.build --diag)
os.environ.setdefault('EVENTCLIP_HIP_LIB', os.path.join(ROOT, 'eventclip_amd', 'libeventclip_hip_diag.so'))
from eventclip_amd import ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
epi = sys.argv[3] if len(sys.argv) > 3 else 'store16'
M = 2560 * 257
A = torch.randn(M, K, device='cuda').half()
W = (torch.randn(N, K, device='cuda') / K ** 0.5).half()
out = torch.zeros(M, N, device='cuda', dtype=torch.float32 if epi.endswith('32') else torch.float16)
tiles = ((M + 255) // 256) * ((N + 255) // 256)
dbg = torch.zeros(tiles * 8 * 2, device='cuda', dtype=torch.float32)
for _ in range(2):
    ops.gemm(A, W, None, epi, out=out, variant=int(os.environ.get('TL_VARIANT', '16')), diag=dbg)
torch.cuda.synchronize()
r = dbg.cpu().numpy().view(np.uint64).reshape(tiles, 8).astype(np.int64)
hw, xcc = r[:, 0], r[:, 6]
cu = ((xcc & 15) << 16) | (hw & 0xff00)       # XCC | SE | SH | CU, dropping wave / simd / pipe ids
print(f'N={N} K={K} {epi}: {tiles} workgroups on {len(np.unique(cu))} CUs')
if os.environ.get('TL_RAW'):
    print('xcc raw', np.unique(xcc)[:20], 'hw fields', [hex(v) for v in np.unique(hw & 0xffff00)[:40]])
    print('first 16 wgs', [(hex(int(a)), int(b)) for a, b in zip(hw[:16], xcc[:16])])
seg = {'prologue (start -> first tiles landed)': r[:, 2] - r[:, 1],
       'main loop': r[:, 3] - r[:, 2],
       'epilogue issue': r[:, 4] - r[:, 3],
       'store drain (vmcnt 0)': r[:, 5] - r[:, 4]}
for k, v in seg.items():
    print(f'  {k:42s} mean {v.mean():9.0f}  p10 {np.percentile(v, 10):9.0f}  p90 {np.percentile(v, 90):9.0f} cycles')
gaps, periods = [], []
for c in np.unique(cu):
    idx = np.where(cu == c)[0]
    idx = idx[np.argsort(r[idx, 1])]
    gaps.append(r[idx[1:], 1] - r[idx[:-1], 5])
    periods.append(np.diff(r[idx, 1]))
gaps, periods = np.concatenate(gaps), np.concatenate(periods)
print(f'  {"gap: stores acknowledged -> next start":42s} mean {gaps.mean():9.0f}  p10 {np.percentile(gaps, 10):9.0f}  '
      f'p90 {np.percentile(gaps, 90):9.0f} cycles')
print(f'  {"period per workgroup on a CU":42s} mean {periods.mean():9.0f}')
# s_memtime is per XCD: measure spans and phase spread inside each XCD
spans, inloop = [], []
for x in np.unique(xcc & 15):
    rx = r[(xcc & 15) == x]
    t0, span = rx[:, 1].min(), rx[:, 5].max() - rx[:, 1].min()
    spans.append(span)
    for t in np.linspace(t0 + span * 0.2, t0 + span * 0.8, 200):
        inloop.append(((rx[:, 2] <= t) & (t < rx[:, 3])).sum())
print(f'  kernel span per XCD: mean {np.mean(spans):.0f} cycles')
print(f'  CUs of an XCD inside the main loop (sampled over the middle 60 %): mean {np.mean(inloop):.1f} of 32, '
      f'min {np.min(inloop)}, max {np.max(inloop)}')
