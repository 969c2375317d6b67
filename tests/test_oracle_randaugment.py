"""The RandAugment oracle against Pillow itself, operator by operator.

The reference applies its operators to PIL images through torchvision 0.13.1's functional_pil
(/root/reference/datasets/augment.py:10-87; torchvision is un-vendored and absent here).  Those
wrappers are one Pillow call each, restated in `pil_op` below with the arguments torchvision passes,
so the oracle is pinned bit for bit against what Pillow computes on event-frame-like and random
images, every magnitude bin, both signs.
"""
import math

import numpy as np
import pytest
from PIL import Image, ImageEnhance, ImageOps

from oracle import randaugment as ora

FILLS = ((255, 255, 255), (0, 0, 0))


def pil_op(img, op, magnitude, fill):
    """What torchvision.transforms.functional does for a PIL image (functional.py / functional_pil.py,
    v0.13.1), called the way augment.py:10-87 calls it (interpolation = BICUBIC, event2img.py:36-42)."""
    w, h = img.size
    opts = dict(fillcolor=tuple(int(v) for v in fill))

    def affine(angle, translate, shear, center):
        if center is None:
            center = [w * 0.5, h * 0.5]
        m = ora.inverse_affine_matrix(center, angle, translate, 1.0, shear)
        return img.transform((w, h), Image.AFFINE, m, Image.BICUBIC, **opts)

    if op == 'Identity':
        return img
    if op == 'ShearX':
        return affine(0.0, [0, 0], [math.degrees(math.atan(magnitude)), 0.0], [0, 0])
    if op == 'ShearY':
        return affine(0.0, [0, 0], [0.0, math.degrees(math.atan(magnitude))], [0, 0])
    if op == 'TranslateX':
        return affine(0.0, [int(magnitude), 0], [0.0, 0.0], None)
    if op == 'TranslateY':
        return affine(0.0, [0, int(magnitude)], [0.0, 0.0], None)
    if op == 'Rotate':
        return img.rotate(magnitude, Image.BICUBIC, False, None, **opts)
    if op == 'Brightness':
        return ImageEnhance.Brightness(img).enhance(1.0 + magnitude)
    if op == 'Color':
        return ImageEnhance.Color(img).enhance(1.0 + magnitude)
    if op == 'Contrast':
        return ImageEnhance.Contrast(img).enhance(1.0 + magnitude)
    if op == 'Sharpness':
        return ImageEnhance.Sharpness(img).enhance(1.0 + magnitude)
    if op == 'Posterize':
        return ImageOps.posterize(img, int(magnitude))
    if op == 'Solarize':
        return ImageOps.solarize(img, magnitude)
    if op == 'AutoContrast':
        return ImageOps.autocontrast(img)
    if op == 'Equalize':
        return ImageOps.equalize(img)
    raise ValueError(op)


def images():
    rng = np.random.default_rng(0)
    out = [rng.integers(0, 256, size=(45, 60, 3), dtype=np.uint8)]
    # an event frame: white background, sparse red / blue pixels of a few intensities
    from oracle import events as oe
    from eventclip_amd.synthetic import make_events
    ev = make_events(9000, (90, 120), seed=3)
    out.append(oe.events2frames(ev, shape=(90, 120), N=9000, grayscale=False, count_non_zero=False,
                                background_mask=True)[0])
    out.append(oe.events2frames(ev, shape=(90, 120), N=9000, grayscale=True, count_non_zero=True,
                                background_mask=False)[0])
    out.append(np.full((20, 33, 3), 77, dtype=np.uint8))          # constant: degenerate histograms
    sq = rng.integers(0, 256, size=(32, 32, 3), dtype=np.uint8)   # square: PIL's rotate fast paths
    out.append(sq)
    return out


@pytest.mark.parametrize('op', ora.OPS)
def test_operator_matches_pillow_on_every_bin(op):
    for img in images():
        H, W, _ = img.shape
        table = ora.magnitude_table(op, (H, W))
        mags = [0.0] if table is None else sorted({float(v) for v in table.tolist()})
        if op in ora.SIGNED:
            mags = mags + [-m for m in mags if m]
        if op == 'Rotate':
            mags += [90.0, 180.0, 270.0, 360.0]
        pil = Image.fromarray(img)
        for fill in FILLS:
            for mag in mags:
                want = np.asarray(pil_op(pil, op, mag, fill))
                got = ora.apply_op(img, op, mag, fill)
                assert got.dtype == np.uint8 and got.shape == img.shape
                if not np.array_equal(got, want):
                    d = np.argwhere(got != want)
                    raise AssertionError(f'{op} magnitude {mag} fill {fill} image {img.shape}: '
                                         f'{len(d)} values differ, first at {d[0].tolist()}: '
                                         f'{got[tuple(d[0])]} vs {want[tuple(d[0])]}')
            if op not in ('ShearX', 'ShearY', 'TranslateX', 'TranslateY', 'Rotate'):
                break                                            # fill only matters for geometry


def test_two_operator_chain_same_ops_for_every_view():
    rng = np.random.default_rng(5)
    frames = rng.integers(0, 256, size=(3, 40, 52, 3), dtype=np.uint8)
    ops = [('Rotate', -17.586206436157227), ('Contrast', 0.34137931466102600)]
    got = ora.randaugment(frames, ops, (255, 255, 255))
    for f, g in zip(frames, got):
        p = Image.fromarray(f)
        for name, mag in ops:
            p = pil_op(p, name, mag, (255, 255, 255))
        np.testing.assert_array_equal(g, np.asarray(p))


def golden():
    import os
    from conftest import GOLDEN
    return np.load(os.path.join(GOLDEN, 'randaugment.npz'))


def test_sampling_matches_the_references_randomize_ops():
    """augment.py:142-157 run by the reference's own class (tools/make_golden_randaugment.py): same
    torch draws, same operator names, same float magnitudes."""
    import torch
    z = golden()
    for shape in ((180, 240), (480, 640)):
        names, mags = z[f'sample_{shape[0]}x{shape[1]}_names'], z[f'sample_{shape[0]}x{shape[1]}_mags']
        for seed in range(len(names)):
            torch.manual_seed(seed)
            ops = ora.randomize_ops(shape)
            assert [o[0] for o in ops] == names[seed].tolist()
            assert [o[1] for o in ops] == mags[seed].tolist()


def test_oracle_matches_the_references_forward():
    """RandAugment.forward of the reference (Pillow standing in for torchvision's PIL branch) on event
    frames: the oracle reproduces every augmented frame bit for bit."""
    z = golden()
    hit = set()
    for tag in z['cases']:
        frames = z[f'g{int(z[tag + "_geo"])}_frames_in']
        ops = list(zip(z[tag + '_op_names'].tolist(), z[tag + '_op_mags'].tolist()))
        got = ora.randaugment(frames, ops, z[tag + '_fill'].tolist())
        np.testing.assert_array_equal(got, z[tag + '_frames_out'], err_msg=f'{tag} {ops}')
        hit |= {o[0] for o in ops}
    assert len(hit) >= 12
