"""Diagnostic: shader clock and socket power while the GEMM runs (run on the GPU box).

Launches GEMMs back to back for a few seconds and polls `rocm-smi` from a child process.
"""
import os
import subprocess
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eventclip_amd import ops  # noqa: E402

M, N, K = 2560 * 257, 3072, 1024
mode = sys.argv[1] if len(sys.argv) > 1 else 'random'
A = torch.randn(M, K, device='cuda').half()
W = (torch.randn(N, K, device='cuda') / K ** 0.5).half()
if mode == 'zeros':
    A.zero_(), W.zero_()
out = torch.empty(M, N, device='cuda', dtype=torch.float16)
print(subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True).stdout[-1500:])
t_end = time.time() + 6
samples = []
n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.time() < t_end:
    for _ in range(50):
        ops.gemm(A, W, None, 'store16', out=out)
        n += 1
    if len(samples) < 4 and time.time() > t_end - 4:
        r = subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True)
        samples.append([ln for ln in r.stdout.splitlines() if 'sclk' in ln or 'Power' in ln or 'mclk' in ln])
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
print(f'{mode}: {ms:.3f} ms per GEMM, {2. * M * N * K / ms / 1e9:.0f} TFLOP/s (includes queue gaps)')
for s in samples:
    print(s)
