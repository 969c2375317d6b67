import os, sys, torch
sys.path.insert(0, '/root/repo')
# EC_ATTN_SPLIT_F32 is read by the DIAGNOSTIC build only since round 6 (the product library's kernel choice never depends on the environment)
os.environ.setdefault('EVENTCLIP_HIP_LIB', os.path.join('/root/repo', 'eventclip_amd', 'libeventclip_hip_diag.so'))
from eventclip_amd import _lib
lib = _lib.lib()
heads, W = 4, 256
for S in (257, 577):
    for scale in (1.5, 4.0, 8.0):
        n = 3
        torch.manual_seed(S)
        qkv = torch.randn(n * S, 3 * W, device='cuda')
        qkv[:, :2 * W] *= scale
        qkv[:, 2 * W:] *= 1.5
        # a few dominant keys per head, placed in both halves, so that the running maximum moves in either pass
        qkv[S // 3::S, W:2 * W] *= 3
        qkv[S - 5::S, W:2 * W] *= 3
        pair = torch.empty((2, n * S, 3 * W), dtype=torch.float16, device='cuda')
        pair[0] = qkv.half(); pair[1] = (qkv - pair[0].float()).half()
        j = pair[0].double() + pair[1].double()
        q, k, v = j.view(n, S, 3, heads, 64).permute(2, 0, 3, 1, 4)
        want = (((q * 0.125) @ k.transpose(-1, -2)).softmax(-1) @ v).permute(0, 2, 1, 3).reshape(n * S, W)
        for f32 in (0, 1):
            if f32: os.environ['EC_ATTN_SPLIT_F32'] = '1'
            else: os.environ.pop('EC_ATTN_SPLIT_F32', None)
            hi = torch.zeros(n * S, W, dtype=torch.float16, device='cuda'); lo = torch.zeros_like(hi)
            _lib.check(lib.ec_attention_split(_lib.ptr(pair[0]), _lib.ptr(pair[1]), _lib.ptr(hi), _lib.ptr(lo), n, S, W, heads, 0, _lib.EC_F16, _lib.stream_ptr()))
            got = hi.double() + lo.double()
            print(S, scale, 'fp32 kernel' if f32 else 'hl kernel  ', 'err %.2e' % float((got - want).abs().max() / want.abs().max()), flush=True)
