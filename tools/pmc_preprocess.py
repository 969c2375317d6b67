"""One geometry of the preprocess kernel for counter collection (tools/pmc_preprocess.sh)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eventclip_amd import preprocess  # noqa: E402
geo = sys.argv[1] if len(sys.argv) > 1 else 'caltech'
shape, n_px, frames = {'caltech': ((180, 240), 224, 2560), 'imagenet': ((480, 640), 224, 1024),
                       'imagenet336': ((480, 640), 336, 1024)}[geo]
fr = torch.randint(0, 256, (frames, *shape, 3), dtype=torch.uint8, device='cuda')
kpad = ((2 * 3 * 14 * 14 + 63) // 64) * 64
out = torch.empty((frames, (n_px // 14) ** 2, kpad), dtype=torch.float16, device='cuda')
for _ in range(3):
    preprocess.preprocess_frames(fr, n_px, 'patches', out=out, patch=14, kpad=kpad)
torch.cuda.synchronize()
