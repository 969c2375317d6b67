"""Race screen of the comparison kernels of the DIAGNOSTIC build (ec_gemm variants 1 / 2 / 3: the plain two-barrier
loop at 256 x 256, 128 x 128 and 128 x 256 tiles).  Round 2 saw the 128 x 128 one return five wrong elements in
3 M once in a dozen test runs; this screen tries to make that happen again: many repeats of the test suite's shapes
with a second stream streaming copies through the memory system WHILE the GEMM runs, every repeat compared bit for
bit with the first, the first with torch.

    python -m eventclip_amd.build --diag && python tools/race_screen_diag.py [repeats] [variants ...]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('EVENTCLIP_HIP_LIB', os.path.join(ROOT, 'eventclip_amd', 'libeventclip_hip_diag.so'))
from eventclip_amd import ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
variants = [int(v) for v in sys.argv[2:]] or [2, 1, 3, 0]
shapes = [(1000, 3072, 1024), (257 * 3, 1024, 1024), (513, 1024, 4096), (77, 768, 640), (5, 512, 64), (256, 256, 128), (300, 48, 64),
          (70001, 768, 640)]
side = torch.cuda.Stream()
noise_a = torch.empty(96 << 20, device='cuda')
noise_b = torch.empty(96 << 20, device='cuda')
total_bad = 0
for v in variants:
    for (M, N, K) in shapes:
        g = torch.Generator(device='cuda').manual_seed(M + N + K)
        A = torch.randn(M, K, device='cuda', generator=g).half()
        W = (torch.randn(N, K, device='cuda', generator=g) / K ** 0.5).half()
        bias = torch.randn(N, device='cuda', generator=g)
        resid = torch.randn(M, N, device='cuda', generator=g)
        want = A.float() @ W.float().t() + bias
        for epi in ('store16', 'resid32', 'gelu16', 'store32'):
            first, bad = None, 0
            for r in range(reps):
                if r % 2 == 0:
                    with torch.cuda.stream(side):       # concurrent traffic, not between launches
                        noise_b.copy_(noise_a)
                if epi == 'resid32':
                    out = resid.clone()
                else:           # poisoned output: a store the kernel misses shows up (a reused buffer would hide it)
                    out = torch.full((M, N), float('nan'), dtype=torch.float16 if epi in ('store16', 'gelu16') else torch.float32,
                                     device='cuda')
                got = ops.gemm(A, W, bias, epi, out=out, variant=v)
                if first is None:
                    first = got.clone()
                    ref = want + resid if epi == 'resid32' else (want * torch.sigmoid(1.702 * want) if epi == 'gelu16' else want)
                    err = float((got.float() - ref).abs().max() / ref.abs().max())
                    if not err <= (2e-3 if epi in ('store16', 'gelu16') else 1e-4):
                        print('MISMATCH vs torch', v, (M, N, K), epi, err, flush=True)
                        bad += 1
                elif not torch.equal(got, first):
                    d = (got != first).nonzero()
                    bad += 1
                    rows = sorted(set(d[:, 0].tolist()))[:6]
                    cols = sorted(set(d[:, 1].tolist()))[:12]
                    print(f'NONDETERMINISTIC variant {v} {(M, N, K)} {epi} run {r}: {d.shape[0]} elements differ; rows {rows} '
                          f'cols {cols} got {got[d[0, 0], d[0, 1]].item()} first {first[d[0, 0], d[0, 1]].item()}', flush=True)
                    if bad >= 3:
                        break
            total_bad += bad
            print(f'variant {v} {(M, N, K)} {epi}: {"CLEAN" if bad == 0 else str(bad) + " BAD"} over {reps} repeats', flush=True)
torch.cuda.synchronize()
print('TOTAL BAD', total_bad)
