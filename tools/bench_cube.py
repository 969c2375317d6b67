"""The GEMM on the square shapes and operand distribution the CDNA programming guide quotes its 256x256 8-phase
template on (4096^3 / 8192^3, uniform [-1, 1) and N(0, 1), bf16 / f16), short bursts at boost clock.
Run on the GPU box: python tools/bench_cube.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eventclip_amd import ops
for n in (4096, 8192):
    for dist in ('uniform', 'normal'):
        for dt in (torch.bfloat16, torch.float16):
            A = (torch.rand(n, n, device='cuda') * 2 - 1 if dist == 'uniform' else torch.randn(n, n, device='cuda')).to(dt)
            W = (torch.rand(n, n, device='cuda') * 2 - 1 if dist == 'uniform' else torch.randn(n, n, device='cuda')).to(dt)
            out = torch.empty(n, n, device='cuda', dtype=dt)
            for v in (0, 4, 5):
                for _ in range(5): ops.gemm(A, W, None, 'store16', out=out, variant=v)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20): ops.gemm(A, W, None, 'store16', out=out, variant=v)
                e1.record(); torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 20
                print(f'{n}^3 {dist:7s} {str(dt)[6:]:8s} variant {v}: {ms:.3f} ms {2*n**3/ms/1e9:.0f} TF', flush=True)
