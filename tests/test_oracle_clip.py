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


# ------------------------------------------------------------------------------------------------
# oracle/clip_ref.py emulate='fp16_reference': the yardstick of the logit-parity tests
# ------------------------------------------------------------------------------------------------
class _LayerNorm16(torch.nn.LayerNorm):
    """openai/CLIP's LayerNorm: fp32 compute on an fp16 tensor, result cast back."""

    def forward(self, x):
        return super().forward(x.type(torch.float32)).type(x.dtype)


class _Block16(torch.nn.Module):
    """ResidualAttentionBlock of the published clip/model.py over torch's own modules."""

    def __init__(self, W, heads):
        super().__init__()
        self.attn = torch.nn.MultiheadAttention(W, heads)
        self.ln_1, self.ln_2 = _LayerNorm16(W), _LayerNorm16(W)
        self.c_fc, self.c_proj = torch.nn.Linear(W, 4 * W), torch.nn.Linear(4 * W, W)

    def forward(self, x):                                     # [S, N, W] fp16
        h = self.ln_1(x)
        x = x + self.attn(h, h, h, need_weights=False)[0]
        h = self.c_fc(self.ln_2(x))
        return x + self.c_proj(h * torch.sigmoid(1.702 * h))


def fp16_emulation_distances(device='cpu', W=128, heads=2, S=10, N=4, L=2, seed=3):
    """(emulation vs torch's half kernels, torch's half kernels vs fp32, emulation vs fp32), max-normalised, for L
    blocks of openai/CLIP's ResidualAttentionBlock built from torch's OWN modules and run in half precision ON
    ``device`` (CPU half kernels here; the MI355X's hipBLASLt / elementwise half kernels in
    tests/test_towers_gpu.py::test_fp16_reference_emulation_against_torch_half_on_the_gpu): nn.MultiheadAttention /
    nn.Linear / elementwise ops on half tensors, LayerNorm computed in fp32, convert_weights' split of what is
    fp16 and what stays fp32."""
    from oracle import clip_ref
    torch.manual_seed(seed)
    blocks = [_Block16(W, heads) for _ in range(L)]
    sd = {}
    for i, b in enumerate(blocks):
        for p in b.parameters():
            p.data.add_(torch.randn_like(p) * 0.05)
        # convert_weights: Linear / MultiheadAttention to fp16, LayerNorm stays fp32
        for mod in (b.attn, b.attn.out_proj, b.c_fc, b.c_proj):
            mod.half()
        pre = f'visual.transformer.resblocks.{i}.'
        sd[pre + 'ln_1.weight'], sd[pre + 'ln_1.bias'] = b.ln_1.weight.data, b.ln_1.bias.data
        sd[pre + 'ln_2.weight'], sd[pre + 'ln_2.bias'] = b.ln_2.weight.data, b.ln_2.bias.data
        sd[pre + 'attn.in_proj_weight'], sd[pre + 'attn.in_proj_bias'] = b.attn.in_proj_weight.data.float(), b.attn.in_proj_bias.data.float()
        sd[pre + 'attn.out_proj.weight'], sd[pre + 'attn.out_proj.bias'] = b.attn.out_proj.weight.data.float(), b.attn.out_proj.bias.data.float()
        sd[pre + 'mlp.c_fc.weight'], sd[pre + 'mlp.c_fc.bias'] = b.c_fc.weight.data.float(), b.c_fc.bias.data.float()
        sd[pre + 'mlp.c_proj.weight'], sd[pre + 'mlp.c_proj.bias'] = b.c_proj.weight.data.float(), b.c_proj.bias.data.float()
    sd = {k: v.clone() for k, v in sd.items()}          # CPU fp32 copies for the oracle, whatever the device
    x = torch.randn(N, S, W)
    with torch.no_grad():
        y = x.half().transpose(0, 1).to(device)
        for b in blocks:
            y = b.to(device)(y)
        y = y.transpose(0, 1).float().cpu()
        emu = clip_ref._blocks_h(clip_ref._h(x), clip_ref.fp16_reference_weights(sd), 'visual.transformer', L, heads)
        exact = clip_ref._blocks(x, sd, 'visual.transformer', L, heads)
    scale = float(exact.abs().max())
    return (float((emu - y).abs().max()) / scale, float((y - exact).abs().max()) / scale,
            float((emu - exact).abs().max()) / scale)


def test_fp16_reference_emulation_matches_torch_half_modules():
    """The emulation (fp32 arithmetic + an explicit round to fp16 after every op) against torch's OWN fp16 kernels
    (CPU half).  Two blocks.  The two are independent fp16 realisations of the same arithmetic (sums associate
    differently, so roundings fall differently): what the yardstick needs is that both sit at the SAME distance from
    fp32 (within 2x) and no further from each other than two such realisations are (< 2x that distance)."""
    d_emu_torch, d_torch_exact, d_emu_exact = fp16_emulation_distances('cpu')
    assert d_emu_torch < 2.0 * d_torch_exact, (d_emu_torch, d_torch_exact)
    assert 0.5 < d_emu_exact / d_torch_exact < 2.0, (d_emu_exact, d_torch_exact)


def test_fp16_reference_emulation_towers_run_and_differ_from_fp32_at_fp16_scale():
    cfg, sd, z = load_tiny()
    from oracle import clip_ref
    img = torch.from_numpy(z['img'].astype(np.float32))
    a = clip_ref.encode_image(sd, cfg, img)
    b = clip_ref.encode_image(sd, cfg, img, emulate='fp16_reference')
    e = float((a - b).abs().max() / a.abs().max())
    assert 1e-5 < e < 2e-2, e
    assert torch.equal(b, b.half().float())                   # an fp16 tensor's values
    tok = torch.from_numpy(z['tok'])
    a, b = clip_ref.encode_text(sd, cfg, tok), clip_ref.encode_text(sd, cfg, tok, emulate='fp16_reference')
    e = float((a - b).abs().max() / a.abs().max())
    assert 1e-5 < e < 2e-2, e
