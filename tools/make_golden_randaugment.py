"""Generate tests/golden/randaugment.npz by running the REFERENCE's own RandAugment class
(/root/reference/datasets/augment.py, build container only).

torchvision is not installed here.  The reference module needs two names from it:
`torchvision.transforms.functional` (F.affine, F.rotate, F.adjust_*, F.posterize, ..., which for PIL
images are one Pillow call each in torchvision 0.13.1's functional_pil) and `InterpolationMode`.
They are stood in by a module that makes exactly those Pillow calls; everything that is the
reference's own logic -- the operator table and magnitudes (`_augmentation_space`), the torch RNG
draws (`randomize_ops`), which arguments each operator passes (`_apply_op`), the same ops for every
view (`forward`) -- runs from the reference's source.  Stored: the inputs (seeded uint8 frames),
the sampled (op, magnitude) lists and the augmented frames.

    python tools/make_golden_randaugment.py
"""
import enum
import importlib.util
import math
import os
import sys
import types

import numpy as np
import torch
from PIL import Image, ImageEnhance, ImageOps

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class InterpolationMode(enum.Enum):
    NEAREST = 'nearest'
    BILINEAR = 'bilinear'
    BICUBIC = 'bicubic'


PIL_MODES = {InterpolationMode.NEAREST: Image.NEAREST, InterpolationMode.BILINEAR: Image.BILINEAR,
             InterpolationMode.BICUBIC: Image.BICUBIC}


def _inverse_affine_matrix(center, angle, translate, scale, shear):
    # torchvision.transforms.functional._get_inverse_affine_matrix, v0.13.1
    rot = math.radians(angle)
    sx, sy = math.radians(shear[0]), math.radians(shear[1])
    cx, cy = center
    tx, ty = translate
    a = math.cos(rot - sy) / math.cos(sy)
    b = -math.cos(rot - sy) * math.tan(sx) / math.cos(sy) - math.sin(rot)
    c = math.sin(rot - sy) / math.cos(sy)
    d = -math.sin(rot - sy) * math.tan(sx) / math.cos(sy) + math.cos(rot)
    m = [d, -b, 0.0, -c, a, 0.0]
    m = [x / scale for x in m]
    m[2] += m[0] * (-cx - tx) + m[1] * (-cy - ty)
    m[5] += m[3] * (-cx - tx) + m[4] * (-cy - ty)
    m[2] += cx
    m[5] += cy
    return m


def _fill(fill):
    return {} if fill is None else {'fillcolor': tuple(int(v) for v in fill)}


class F:   # the functions augment.py calls, PIL branch of torchvision 0.13.1
    @staticmethod
    def get_dimensions(img):
        return [len(img.getbands()), img.size[1], img.size[0]]

    @staticmethod
    def affine(img, angle, translate, scale, shear, interpolation=InterpolationMode.NEAREST, fill=None,
               center=None):
        w, h = img.size
        if center is None:
            center = [w * 0.5, h * 0.5]
        m = _inverse_affine_matrix(center, angle, translate, scale, shear)
        return img.transform((w, h), Image.AFFINE, m, PIL_MODES[interpolation], **_fill(fill))

    @staticmethod
    def rotate(img, angle, interpolation=InterpolationMode.NEAREST, expand=False, center=None, fill=None):
        return img.rotate(angle, PIL_MODES[interpolation], expand, center, **_fill(fill))

    adjust_brightness = staticmethod(lambda img, f: ImageEnhance.Brightness(img).enhance(f))
    adjust_saturation = staticmethod(lambda img, f: ImageEnhance.Color(img).enhance(f))
    adjust_contrast = staticmethod(lambda img, f: ImageEnhance.Contrast(img).enhance(f))
    adjust_sharpness = staticmethod(lambda img, f: ImageEnhance.Sharpness(img).enhance(f))
    posterize = staticmethod(lambda img, bits: ImageOps.posterize(img, bits))
    solarize = staticmethod(lambda img, thr: ImageOps.solarize(img, thr))
    autocontrast = staticmethod(lambda img: ImageOps.autocontrast(img))
    equalize = staticmethod(lambda img: ImageOps.equalize(img))
    invert = staticmethod(lambda img: ImageOps.invert(img))


def load_reference():
    tv = types.ModuleType('torchvision')
    tvt = types.ModuleType('torchvision.transforms')
    tvt.functional = F
    tvt.InterpolationMode = InterpolationMode
    tv.transforms = tvt
    sys.modules.update({'torchvision': tv, 'torchvision.transforms': tvt,
                        'torchvision.transforms.functional': F})
    spec = importlib.util.spec_from_file_location('ref_augment', '/root/reference/datasets/augment.py')
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ref = load_reference()
    from eventclip_amd.synthetic import make_events
    from oracle import events as oe
    out = {}
    cases = []
    # event frames of two sensor geometries (white / black background: event2img.py:36-42 picks the fill)
    geos = [((180, 240), 20000, False, True, [255, 255, 255]), ((100, 120), 30000, True, False, [0, 0, 0])]
    ci = 0
    for (shape, N, gray, bg, fill) in geos:
        ev = make_events(int(2.6 * N), shape, seed=11 + ci)
        frames = oe.events2frames(ev, shape=shape, N=N, grayscale=gray, count_non_zero=not bg,
                                  background_mask=bg)
        for seed in range(12):
            aug = ref.RandAugment(num_ops=2, interpolation=InterpolationMode.BICUBIC, fill=fill)
            torch.manual_seed(1000 * ci + seed)
            # peek at the draws the forward pass is about to make (same seed -> same ops)
            aug.randomize_ops(shape)
            ops = list(aug.cur_ops)
            aug.cur_ops = None
            torch.manual_seed(1000 * ci + seed)
            res = aug([Image.fromarray(f) for f in frames])
            res = np.stack([np.asarray(r) for r in res])
            tag = f'c{len(cases)}'
            out[f'g{ci}_frames_in'] = frames
            out[tag + '_geo'] = ci
            out[tag + '_frames_out'] = res
            out[tag + '_op_names'] = np.array([o[0] for o in ops])
            out[tag + '_op_mags'] = np.array([o[1] for o in ops], dtype=np.float64)
            out[tag + '_fill'] = np.array(fill)
            out[tag + '_seed'] = 1000 * ci + seed
            cases.append(tag)
        ci += 1
    # the sampling alone over many seeds and both image sizes
    for shape in ((180, 240), (480, 640)):
        names, mags = [], []
        for seed in range(200):
            aug = ref.RandAugment(num_ops=2, interpolation=InterpolationMode.BICUBIC, fill=None)
            torch.manual_seed(seed)
            aug.randomize_ops(shape)
            names.append([o[0] for o in aug.cur_ops])
            mags.append([o[1] for o in aug.cur_ops])
        out[f'sample_{shape[0]}x{shape[1]}_names'] = np.array(names)
        out[f'sample_{shape[0]}x{shape[1]}_mags'] = np.array(mags, dtype=np.float64)
    out['cases'] = np.array(cases)
    path = os.path.join(ROOT, 'tests', 'golden', 'randaugment.npz')
    np.savez_compressed(path, **out)
    used = sorted({str(n) for c in cases for n in out[c + '_op_names']})
    print(f'wrote {path}: {len(cases)} cases, {os.path.getsize(path) / 1024:.0f} KiB; operators hit: {used}')


if __name__ == '__main__':
    main()
