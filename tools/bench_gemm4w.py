"""The four-wave probe kernel (diagnostic build, variant 20) against the default kernel: correctness against torch and
ms per launch on the tower's shapes (16-bit store epilogue).  python -m eventclip_amd.build --diag first."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('EVENTCLIP_HIP_LIB', os.path.join(ROOT, 'eventclip_amd', 'libeventclip_hip_diag.so'))
from eventclip_amd import ops  # noqa: E402

M = 2560 * 257
for name, N, K in (('QKV', 3072, 1024), ('out_proj', 1024, 1024), ('c_fc', 4096, 1024), ('c_proj', 1024, 4096)):
    g = torch.Generator(device='cuda').manual_seed(N + K)
    A = torch.randn(M, K, device='cuda', generator=g).half()
    W = (torch.randn(N, K, device='cuda', generator=g) / K ** 0.5).half()
    bias = torch.randn(N, device='cuda', generator=g)
    out = {v: torch.empty(M, N, dtype=torch.float16, device='cuda') for v in (0, 20, 21)}
    for v in (0, 20, 21):
        ops.gemm(A, W, bias, 'store16', out=out[v], variant=v)
    torch.cuda.synchronize()
    ref = (A[:4096].float() @ W.float().t() + bias)
    e0 = float((out[0][:4096].float() - ref).abs().max() / ref.abs().max())
    e20 = float((out[20][:4096].float() - ref).abs().max() / ref.abs().max())
    e21 = float((out[21][:4096].float() - ref).abs().max() / ref.abs().max())
    same = (torch.equal(out[0], out[20]), torch.equal(out[0], out[21]))
    times = {0: [], 20: [], 21: []}
    for _ in range(5):
        for v in (0, 20, 21):
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a0.record()
            for _ in range(5):
                ops.gemm(A, W, bias, 'store16', out=out[v], variant=v)
            a1.record()
            torch.cuda.synchronize()
            times[v].append(a0.elapsed_time(a1) / 5)
    t0, t20, t21 = sorted(times[0])[2], sorted(times[20])[2], sorted(times[21])[2]
    print(f'{name:9s} N={N} K={K}: default {t0:.3f} ms = {2.0 * M * N * K / t0 / 1e9:5.0f} TFLOP/s (err {e0:.1e}) | four-wave probe {t20:.3f} ms = '
          f'{2.0 * M * N * K / t20 / 1e9:5.0f} TFLOP/s | interleaved quarter ring {t21:.3f} ms = {2.0 * M * N * K / t21 / 1e9:5.0f} TFLOP/s (err {e20:.1e} / {e21:.1e}, bit-identical to default: {same})', flush=True)
    del A, W, out
