import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from eventclip_amd import ops
M, N, K = 657920, 3072, 1024
A = torch.randn(M, K, device='cuda').half()
W = (torch.randn(N, K, device='cuda') / 32).half()
bias = torch.randn(N, device='cuda')
out = torch.empty(M, N, dtype=torch.float16, device='cuda')
st = ops.row_stats(A); cs = W.float().sum(1).contiguous()
cases = {'store16 bias': lambda: ops.gemm(A, W, bias, 'store16', out=out),
         'store16 nobias': lambda: ops.gemm(A, W, None, 'store16', out=out),
         'store16_ln': lambda: ops.gemm(A, W, bias, 'store16_ln', out=out, row_stats=st, col_sums=cs),
         'gelu16 bias': lambda: ops.gemm(A, W, bias, 'gelu16', out=out),
         'gelu16_ln': lambda: ops.gemm(A, W, bias, 'gelu16_ln', out=out, row_stats=st, col_sums=cs)}
for f in cases.values(): f(); f()
torch.cuda.synchronize()
res = {k: [] for k in cases}
for r in range(4):
    for k, f in cases.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(6): f()
        e1.record(); torch.cuda.synchronize()
        res[k].append(e0.elapsed_time(e1) / 6)
for k, v in res.items(): print(f'{k:16s}', ' '.join(f'{x:.3f}' for x in v))
