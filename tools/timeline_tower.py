"""Diagnostic: per-tile timeline of the PERSISTENT default GEMM (variant 18 of the diagnostic build) on the tower's four
shapes with the epilogues the tower actually runs (store16_ln, resid_hl + row sums, gelu16_ln), wave 0 of every
workgroup, cycles per 256 x 256 tile.  Run on the GPU box.

    python tools/timeline_tower.py [frames]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('EVENTCLIP_HIP_LIB', os.path.join(ROOT, 'eventclip_amd', 'libeventclip_hip_diag.so'))
from eventclip_amd import ops  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 2560
M = frames * 257
VAR = int(os.environ.get('TL_VARIANT', '18'))


def run(name, N, K, epi):
    g = torch.Generator(device='cuda').manual_seed(N + K)
    A = torch.randn(M, K, device='cuda', generator=g).half()
    W = (torch.randn(N, K, device='cuda', generator=g) / K ** 0.5).half()
    bias = torch.randn(N, device='cuda', generator=g)
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    dbg = torch.zeros(tiles * 8 * 2, device='cuda', dtype=torch.float32)
    kw = {}
    if epi == 'resid_hl':
        out = torch.randn(M, N, device='cuda', generator=g).half()
        kw = dict(aux=torch.zeros(M, N, device='cuda', dtype=torch.float16),
                  row_sums=torch.zeros(M, N // 64, 2, device='cuda'))
    else:
        out = torch.empty(M, N, device='cuda', dtype=torch.float16)
        if epi.endswith('_ln'):
            kw = dict(row_stats=ops.row_stats(A), col_sums=W.float().sum(1).contiguous())
    # clock: a few un-stamped launches first so the chip is at its sustained state
    for _ in range(6):
        ops.gemm(A, W, bias, epi, out=out, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        ops.gemm(A, W, bias, epi, out=out, **kw)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 4
    ops.gemm(A, W, bias, epi, out=out, variant=VAR, diag=dbg, **kw)
    torch.cuda.synchronize()
    r = dbg.cpu().numpy().view(np.uint64).reshape(tiles, 8).astype(np.int64)
    # tiles a workgroup walked: id, id + 256, ...  (records are indexed by the launch-order tile id)
    nwg = min(tiles, 256)
    per_wg = [r[b::nwg] for b in range(nwg)]
    seg = {k: [] for k in ('prologue (turn-over barrier -> loop)', 'main loop', 'epilogue issue', 'store drain (vmcnt 0)',
                           'hand-over barrier', 'period')}
    for t in per_wg:
        if len(t) < 4:
            continue
        mid = t[1:-1]             # steady state: not the first tile (cold prologue), not the last
        nxt = t[2:]
        seg['prologue (turn-over barrier -> loop)'].append(mid[:, 2] - mid[:, 1])
        seg['main loop'].append(mid[:, 3] - mid[:, 2])
        seg['epilogue issue'].append(mid[:, 4] - mid[:, 3])
        seg['store drain (vmcnt 0)'].append(mid[:, 5] - mid[:, 4])
        seg['hand-over barrier'].append(nxt[:, 1] - mid[:, 5])
        seg['period'].append(nxt[:, 1] - mid[:, 1])
    nk = K // 64
    print(f'{name}: N={N} K={K} {epi}: {ms:.3f} ms un-stamped = {2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s; {tiles} tiles', flush=True)
    period = np.concatenate(seg['period']).mean()
    for k, v in seg.items():
        v = np.concatenate(v)
        extra = f'   ({v.mean() / nk:.0f} per K tile; MFMA time 2048)' if k == 'main loop' else ''
        print(f'  {k:40s} mean {v.mean():9.0f}  p10 {np.percentile(v, 10):9.0f}  p90 {np.percentile(v, 90):9.0f}  '
              f'{100 * v.mean() / period:5.1f} %{extra}')
    # do the workgroups stay in step?  epilogue + hand-over by round (tile id // workgroups), and the spread of the
    # epilogues' start times inside a round, in tile periods
    rounds = tiles // nwg
    rr = r[:rounds * nwg].reshape(rounds, nwg, 8)
    t0 = rr[:, :, 1].min()
    print('  round: epilogue issue + drain / start-time spread (std, in periods) / epilogue start, first wg (cycles since launch)')
    for k in sorted(set([1, 2, 3, 5, 8, 12, 20, 30, rounds - 2])):
        if 1 <= k < rounds - 1:
            epi = (rr[k, :, 5] - rr[k, :, 3]).mean()
            start = rr[k, :, 3]
            print(f'    {k:3d}: {epi:8.0f}   {start.std() / period:5.2f}   {start.min() - t0:10d}')
    del A, W, out, dbg


for name, N, K, epi in (('QKV', 3072, 1024, 'store16_ln'), ('out_proj', 1024, 1024, 'resid_hl'),
                        ('c_fc', 4096, 1024, 'gelu16_ln'), ('c_proj', 1024, 4096, 'resid_hl'),
                        ('QKV plain', 3072, 1024, 'store16')):
    run(name, N, K, epi)
