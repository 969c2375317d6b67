"""What one wave per SIMD can issue (diagnostic build): cycles per block of 64 independent v_mfma_f32_16x16x32_f16 on AGPR
accumulators, alone and with what the four-wave GEMM layout puts between blocks (profiles/r3_gemm.md 7).

    python -m eventclip_amd.build --diag && python tools/mfma_probe.py
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('EVENTCLIP_HIP_LIB', os.path.join(ROOT, 'eventclip_amd', 'libeventclip_hip_diag.so'))
from eventclip_amd import _lib  # noqa: E402

h = ctypes.CDLL(os.environ['EVENTCLIP_HIP_LIB'])
h.ec_mfma_probe.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
_lib.require_gpu()
src = torch.zeros(64 << 20, dtype=torch.uint8, device='cuda')
out = torch.zeros(1024, device='cuda')
iters, wgs = 4000, 256
names = {0: '64 MFMAs', 1: '+ 16 ds_read_b128', 2: '+ barrier', 3: '+ reads + barrier', 4: '+ 8 LDS-DMA', 5: '+ reads + DMA',
         6: '+ barrier + DMA', 7: '+ reads + barrier + DMA (the GEMM quarter)'}
names.update({9: '+ reads SPREAD through the block', 11: '+ spread reads + barrier', 13: '+ spread reads + DMA', 15: '+ spread reads + barrier + DMA'})
for mode in (0, 1, 2, 3, 4, 5, 6, 7, 9, 11, 13, 15):
    for _ in range(2):
        assert h.ec_mfma_probe(mode, iters, wgs, _lib.ptr(src), _lib.ptr(out), _lib.stream_ptr()) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    assert h.ec_mfma_probe(mode, iters, wgs, _lib.ptr(src), _lib.ptr(out), _lib.stream_ptr()) == 0
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    tf = 2.0 * 16 * 16 * 32 * 64 * 4 * wgs * iters / ms / 1e9
    print(f'mode {mode} {names[mode]:44s}: {ms:8.3f} ms  {ms * 1e6 / iters:8.1f} ns per block  {tf:6.0f} TFLOP/s', flush=True)
