"""Full-depth tower outputs of the CPU oracle for seeded random weights and inputs ->
tests/golden/towers_seeded.npz.  The GPU tests regenerate the same weights/inputs from
the seeds (eventclip_amd.clip.random_state_dict, torch.Generator) and compare against
these stored oracle outputs instead of re-running a 160-GFLOP/image fp32 model on the
GPU box's host.

    python tools/make_golden_vit.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from eventclip_amd import clip as eclip  # noqa: E402
from oracle import clip_ref  # noqa: E402

CASES = [  # name, arch, overrides, weight seed, input seed, n
    ('vitl14', 'ViT-L/14', dict(text_layers=1, vocab_size=1024), 11, 5, 2),
    ('vitb32', 'ViT-B/32', dict(text_layers=1, vocab_size=1024), 11, 5, 3),
    ('vitl14_336', 'ViT-L/14@336px', dict(layers=4, text_layers=1, vocab_size=1024), 11, 5, 1),
]


def seeded_images(n, R, seed):
    return torch.randn(n, 3, R, R, generator=torch.Generator().manual_seed(seed))


def main():
    torch.set_num_threads(os.cpu_count() or 8)
    out = {}
    for name, arch, ov, wseed, iseed, n in CASES:
        cfg = eclip.arch_config(arch, **ov)
        sd = eclip.random_state_dict(cfg, seed=wseed)
        img = seeded_images(n, cfg['image_size'], iseed)
        out[name] = clip_ref.encode_image(sd, cfg, img).numpy()
        out[name + '_w16'] = clip_ref.encode_image(clip_ref.round_weights(sd, torch.float16), cfg,
                                                   img).numpy()
        out[name + '_img_checksum'] = np.array(float(img.double().sum()))
        print(name, out[name].shape, float(np.abs(out[name]).max()))
        # the same weights and images through HF transformers' CLIP vision tower at the full geometry
        # (and depth, except the 336-px case): the restatement is not only checked on a toy model
        import make_golden_clip as mgc
        hf = mgc.hf_model(cfg, {**sd, 'logit_scale': torch.tensor(float(np.log(100.0)))})
        with torch.no_grad():
            hf_img = mgc.feats(hf.get_image_features(pixel_values=img))
        rel = float((torch.from_numpy(out[name]) - hf_img).abs().max() / hf_img.abs().max())
        print(f'  oracle vs HF transformers at {arch} {ov}: rel {rel:.2e}')
        assert rel < 2e-5, rel
        out[name + '_hf'] = hf_img.numpy()
        del hf
    # text towers, full depth
    for name, arch in (('text_l14', 'ViT-L/14'), ('text_b32', 'ViT-B/32')):
        cfg = eclip.arch_config(arch, layers=1)
        sd = eclip.random_state_dict(cfg, seed=12)
        tok = eclip.synthetic_tokens(9, seed=3)
        out[name] = clip_ref.encode_text(sd, cfg, tok).numpy()
        print(name, out[name].shape)
        import make_golden_clip as mgc
        hf = mgc.hf_model(cfg, {**sd, 'logit_scale': torch.tensor(float(np.log(100.0)))})
        with torch.no_grad():
            hf_txt = mgc.feats(hf.get_text_features(input_ids=tok.long()))
        rel = float((torch.from_numpy(out[name]) - hf_txt).abs().max() / hf_txt.abs().max())
        print(f'  oracle vs HF transformers text tower of {arch}: rel {rel:.2e}')
        assert rel < 2e-5, rel
        out[name + '_hf'] = hf_txt.numpy()
        del hf
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'towers_seeded.npz'), **out)


if __name__ == '__main__':
    main()
