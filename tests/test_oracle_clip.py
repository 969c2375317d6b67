"""CLIP oracle against the HF-transformers outputs stored in tests/golden/clip_tiny.npz,
and the preprocess oracle against Pillow itself."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN


def load_tiny():
    z = np.load(os.path.join(GOLDEN, 'clip_tiny.npz'))
    cfg = {str(k): int(v) for k, v in zip(z['cfg_keys'], z['cfg_vals'])}
    sd = {k[2:]: torch.from_numpy(z[k].astype(np.float32)) for k in z.files if k.startswith('w:')}
    sd['logit_scale'] = torch.tensor(float(np.log(100.0)))
    return cfg, sd, z


def test_oracle_towers_match_hf_fixture():
    from oracle import clip_ref
    cfg, sd, z = load_tiny()
    img = torch.from_numpy(z['img'].astype(np.float32))
    tok = torch.from_numpy(z['tok'])
    oi = clip_ref.encode_image(sd, cfg, img).numpy()
    ot = clip_ref.encode_text(sd, cfg, tok).numpy()
    assert np.abs(oi - z['hf_img']).max() / np.abs(z['hf_img']).max() < 1e-5
    assert np.abs(ot - z['hf_txt']).max() / np.abs(z['hf_txt']).max() < 1e-5


@pytest.mark.parametrize('shape,n_px', [((180, 240), 224), ((100, 120), 224), ((480, 640), 224),
                                        ((480, 640), 336), ((224, 224), 224), ((300, 224), 224)])
def test_preprocess_oracle_matches_pillow(shape, n_px):
    from PIL import Image
    from oracle import preprocess as op
    rng = np.random.default_rng(shape[0] + n_px)
    img = rng.integers(0, 256, size=(*shape, 3), dtype=np.uint8)
    nh, nw = op.resized_size(*shape, n_px)
    want = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BICUBIC))
    np.testing.assert_array_equal(op.resize_bicubic(img, nh, nw), want)
    top, left = op.center_crop_offsets(nh, nw, n_px)
    crop = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BICUBIC).crop(
        (left, top, left + n_px, top + n_px)))
    np.testing.assert_array_equal(op.resize_crop_u8(img[None], n_px)[0], crop)


def test_preprocess_oracle_normalise_matches_torch_ops():
    from oracle import preprocess as op
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(1, 100, 120, 3), dtype=np.uint8)
    got = op.preprocess(img, 224)
    u8 = torch.from_numpy(op.resize_crop_u8(img, 224)[0])
    t = u8.permute(2, 0, 1).to(torch.float32).div(255)          # ToTensor
    mean = torch.tensor(op.CLIP_MEAN).view(3, 1, 1)
    std = torch.tensor(op.CLIP_STD).view(3, 1, 1)
    np.testing.assert_array_equal(got[0], ((t - mean) / std).numpy())   # Normalize


def test_patchify_layout_matches_conv():
    from oracle import preprocess as op
    import torch.nn.functional as F
    x = torch.randn(2, 3, 28, 28)
    w = torch.randn(8, 3, 14, 14)
    p = torch.from_numpy(op.patchify(x.numpy(), 14))
    wp = torch.zeros(8, p.shape[-1])
    wp[:, :588] = w.reshape(8, -1)
    got = p @ wp.t()                                             # [2, 4, 8]
    want = F.conv2d(x, w, stride=14).reshape(2, 8, 4).permute(0, 2, 1)
    torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-4)


def test_full_size_oracle_outputs_agree_with_hf_transformers():
    """tests/golden/towers_seeded.npz holds, for ViT-L/14 and ViT-B/32 at full depth (and L/14@336px at 4
    layers) and the full-depth text towers, both the oracle's features and HF transformers' for the same seeded
    weights and inputs."""
    import os
    import numpy as np
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, 'towers_seeded.npz'))
    for name in ('vitl14', 'vitb32', 'vitl14_336', 'text_l14', 'text_b32'):
        a, b = z[name], z[name + '_hf']
        assert a.shape == b.shape
        assert np.abs(a - b).max() / np.abs(b).max() < 2e-5
