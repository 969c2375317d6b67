import os, sys, torch
sys.path.insert(0, os.getcwd())
from eventclip_amd import ops
for name, N, K, epi in (('out', 1024, 1024, 'resid32'), ('fc2', 1024, 4096, 'resid32'), ('qkv', 3072, 1024, 'store16'), ('fc1', 4096, 1024, 'gelu16'), ('dh_fc1T', 1024, 4096, 'store32'), ('dh_qkvT', 1024, 3072, 'store32')):
    for M in (16384, 16448, 16640, 32768):
        A = torch.randn(M, K, device='cuda').half(); W = (torch.randn(N, K, device='cuda') / K ** 0.5).half()
        out = torch.zeros(M, N, device='cuda', dtype=torch.float32 if epi.endswith('32') else torch.float16)
        for _ in range(3): ops.gemm(A, W, None, epi, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.gemm(A, W, None, epi, out=out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f'{name} M={M} {ms*1e3:.1f} us {2*M*N*K/ms/1e9:.0f} TF', flush=True)
