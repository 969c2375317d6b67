"""Oracle for the frame-space RandAugment of the reference (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates /root/reference/datasets/augment.py: `_apply_op` (:10-87, the 14 operators of
`_augmentation_space`, :123-140) on uint8 RGB frames, and `randomize_ops` (:142-157, the torch RNG
draws).  The reference runs the operators on PIL images through torchvision 0.13.1's functional_pil
(un-vendored; thin wrappers over Pillow): F.affine / F.rotate -> Image.transform(AFFINE, BICUBIC,
fillcolor), adjust_* -> ImageEnhance, posterize / solarize / autocontrast / equalize -> ImageOps.
torchvision is absent from this image but Pillow is not, so every operator here is pinned LIVE
against the Pillow call chain torchvision makes (tests/test_oracle_randaugment.py), bit for bit;
the sampling is pinned by a fixture produced by the reference's own RandAugment class
(tools/make_golden_randaugment.py).

All functions take and return uint8 [H, W, 3] numpy arrays.
"""
import math

import numpy as np

OPS = ('Identity', 'ShearX', 'ShearY', 'TranslateX', 'TranslateY', 'Rotate', 'Brightness', 'Color',
       'Contrast', 'Sharpness', 'Posterize', 'Solarize', 'AutoContrast', 'Equalize')
SIGNED = {'ShearX', 'ShearY', 'TranslateX', 'TranslateY', 'Rotate', 'Brightness', 'Color', 'Contrast',
          'Sharpness'}
NUM_BINS = 30


# ---------------------------------------------------------------------------------------------
# sampling (augment.py:123-157)
# ---------------------------------------------------------------------------------------------
def magnitude_table(op, image_size):
    """float magnitudes per bin, or None for the operators without one (augment.py:123-140)."""
    import torch
    H, W = image_size
    n = NUM_BINS
    if op in ('ShearX', 'ShearY'):
        return torch.linspace(0.0, 0.3, n)
    if op == 'TranslateX':
        return torch.linspace(0.0, 150.0 / 331.0 * W, n)
    if op == 'TranslateY':
        return torch.linspace(0.0, 150.0 / 331.0 * H, n)
    if op == 'Rotate':
        return torch.linspace(0.0, 30.0, n)
    if op in ('Brightness', 'Color', 'Contrast', 'Sharpness'):
        return torch.linspace(0.0, 0.9, n)
    if op == 'Posterize':
        return 8 - (torch.arange(n) / ((n - 1) / 4)).round().int()
    if op == 'Solarize':
        return torch.linspace(255.0, 0.0, n)
    return None


def randomize_ops(image_size, num_ops=2, generator=None):
    """augment.py:142-157 with the same torch draws in the same order -> [(op_name, magnitude)]."""
    import torch
    kw = {} if generator is None else {'generator': generator}
    cur = int(torch.randint(NUM_BINS, (1,), **kw).item())
    ops = []
    for _ in range(num_ops):
        name = OPS[int(torch.randint(len(OPS), (1,), **kw).item())]
        table = magnitude_table(name, image_size)
        mag = float(table[cur].item()) if table is not None else 0.0
        if name in SIGNED and int(torch.randint(2, (1,), **kw)):
            mag *= -1.0
        ops.append((name, mag))
    return ops


# ---------------------------------------------------------------------------------------------
# geometry: torchvision F.affine / F.rotate on PIL images
# ---------------------------------------------------------------------------------------------
def inverse_affine_matrix(center, angle, translate, scale, shear):
    """torchvision.transforms.functional._get_inverse_affine_matrix (0.13.1): output -> input."""
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


def rotate_matrix(angle, size):
    """PIL.Image.Image.rotate's matrix (expand=False, center=None, translate=None); None for the
    fast paths that return the image unchanged or transposed."""
    w, h = size
    angle = angle % 360.0
    if angle == 0:
        return 'copy'
    if angle == 180:
        return 'rot180'
    if angle in (90, 270) and w == h:
        return 'rot90' if angle == 90 else 'rot270'
    cx, cy = w / 2.0, h / 2.0
    ang = -math.radians(angle)
    m = [round(math.cos(ang), 15), round(math.sin(ang), 15), 0.0,
         round(-math.sin(ang), 15), round(math.cos(ang), 15), 0.0]
    m[2] = m[0] * (-cx) + m[1] * (-cy) + m[2]
    m[5] = m[3] * (-cx) + m[4] * (-cy) + m[5]
    m[2] += cx
    m[5] += cy
    return m


def op_matrix(op, magnitude, size):
    """The 6 coefficients Image.transform(AFFINE) receives for a geometric operator, or a string for
    PIL's rotate fast paths.  size = (W, H)."""
    w, h = size
    if op == 'ShearX':
        return inverse_affine_matrix([0, 0], 0.0, [0, 0], 1.0, [math.degrees(math.atan(magnitude)), 0.0])
    if op == 'ShearY':
        return inverse_affine_matrix([0, 0], 0.0, [0, 0], 1.0, [0.0, math.degrees(math.atan(magnitude))])
    if op == 'TranslateX':
        return inverse_affine_matrix([w * 0.5, h * 0.5], 0.0, [int(magnitude), 0], 1.0, [0.0, 0.0])
    if op == 'TranslateY':
        return inverse_affine_matrix([w * 0.5, h * 0.5], 0.0, [0, int(magnitude)], 1.0, [0.0, 0.0])
    if op == 'Rotate':
        return rotate_matrix(magnitude, size)
    raise ValueError(op)


def affine_bicubic(img, m, fill):
    """Pillow's Image.transform(size, AFFINE, m, BICUBIC, fillcolor=fill) for an RGB image
    (libImaging/Geometry.c: ImagingGenericTransform + affine_transform + bicubic_filter32RGB), in
    float64 with Pillow's operation order."""
    H, W, _ = img.shape
    a0, a1, a2, a3, a4, a5 = [float(v) for v in m]
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing='ij')
    xin0 = a0 * (xs + 0.5) + a1 * (ys + 0.5) + a2
    yin0 = a3 * (xs + 0.5) + a4 * (ys + 0.5) + a5
    inside = (xin0 >= 0.0) & (xin0 < W) & (yin0 >= 0.0) & (yin0 < H)
    xin = xin0 - 0.5
    yin = yin0 - 0.5
    fx = np.floor(xin)
    fy = np.floor(yin)
    dx = xin - fx
    dy = yin - fy
    x = fx.astype(np.int64) - 1
    y = fy.astype(np.int64) - 1
    src = img.astype(np.float64)

    def cubic(v1, v2, v3, v4, d):
        p1 = v2
        p2 = -v1 + v3
        p3 = 2 * (v1 - v2) + v3 - v4
        p4 = -v1 + v2 - v3 + v4
        return p1 + d * (p2 + d * (p3 + d * p4))

    xc = [np.clip(x + k, 0, W - 1) for k in range(4)]
    out = np.empty_like(img)
    for c in range(3):
        ch = src[:, :, c]

        def row(yy):
            return cubic(ch[yy, xc[0]], ch[yy, xc[1]], ch[yy, xc[2]], ch[yy, xc[3]], dx)

        v1 = row(np.clip(y, 0, H - 1))
        rows = [v1]
        for k in (1, 2, 3):
            ok = (y + k >= 0) & (y + k < H)
            vk = row(np.clip(y + k, 0, H - 1))
            rows.append(np.where(ok, vk, rows[-1]))
        v = cubic(rows[0], rows[1], rows[2], rows[3], dy)
        q = np.where(v <= 0.0, 0, np.where(v >= 255.0, 255, np.floor(v))).astype(np.uint8)   # (UINT8) v: truncation
        out[:, :, c] = np.where(inside, q, np.uint8(fill[c]))
    return out


# ---------------------------------------------------------------------------------------------
# colour: ImageEnhance = Image.blend(degenerate, image, factor)
# ---------------------------------------------------------------------------------------------
def to_L(img):
    """Pillow's RGB -> L: (R * 19595 + G * 38470 + B * 7471 + 0x8000) >> 16."""
    r, g, b = (img[:, :, k].astype(np.int64) for k in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def blend(deg, img, alpha):
    """libImaging/Blend.c: out = in1 + alpha * (in2 - in1) in C float, truncated (clipped when the
    factor extrapolates)."""
    alpha = np.float32(alpha)
    if alpha == np.float32(0.0):
        return deg.copy()
    if alpha == np.float32(1.0):
        return img.copy()
    a = deg.astype(np.int32)
    d = img.astype(np.int32) - a
    t = a.astype(np.float32) + alpha * d.astype(np.float32)       # float: one rounding per operation
    if 0.0 <= alpha <= 1.0:
        return t.astype(np.int32).astype(np.uint8)
    return np.where(t <= 0.0, 0, np.where(t >= 255.0, 255, t.astype(np.int32))).astype(np.uint8)


def smooth(img):
    """ImageFilter.SMOOTH: 3x3 kernel (1 1 1 / 1 5 1 / 1 1 1) / 13, border pixels copied
    (libImaging/Filter.c, float accumulation, + 0.5 and truncation)."""
    k = np.array([1, 1, 1, 1, 5, 1, 1, 1, 1], dtype=np.float32) / np.float32(13)
    H, W, _ = img.shape
    out = img.copy()
    src = img.astype(np.float32)
    acc = np.full((H - 2, W - 2, 3), np.float32(0.5), dtype=np.float32)
    # Filter.c sums the three kernel rows bottom-up is irrelevant for bit-exactness only if the
    # order matches: ss = offset; ss += row(y+1) k[0..2]; ss += row(y) k[3..5]; ss += row(y-1) k[6..8]
    order = [(2, 0), (1, 3), (0, 6)]
    for dy, k0 in order:
        rows = src[dy:dy + H - 2]
        part = rows[:, 0:W - 2] * k[k0] + rows[:, 1:W - 1] * k[k0 + 1] + rows[:, 2:W] * k[k0 + 2]
        acc = acc + part
    q = np.where(acc <= 0.0, 0, np.where(acc >= 255.0, 255, acc.astype(np.int32))).astype(np.uint8)
    out[1:H - 1, 1:W - 1] = q
    return out


def lut_apply(img, lut):
    """lut: uint8 [3, 256] (per band, ImageOps._lut with a 768-entry table)."""
    out = np.empty_like(img)
    for c in range(3):
        out[:, :, c] = lut[c][img[:, :, c]]
    return out


def autocontrast_lut(img):
    lut = np.zeros((3, 256), dtype=np.uint8)
    for c in range(3):
        h = np.bincount(img[:, :, c].ravel(), minlength=256)
        nz = np.flatnonzero(h)
        lo, hi = int(nz[0]), int(nz[-1])
        if hi <= lo:
            lut[c] = np.arange(256)
            continue
        scale = 255.0 / (hi - lo)
        offset = -lo * scale
        for ix in range(256):
            v = int(ix * scale + offset)
            lut[c, ix] = 0 if v < 0 else (255 if v > 255 else v)
    return lut


def equalize_lut(img):
    lut = np.zeros((3, 256), dtype=np.uint8)
    for c in range(3):
        h = np.bincount(img[:, :, c].ravel(), minlength=256)
        histo = h[h > 0]
        if len(histo) <= 1:
            lut[c] = np.arange(256)
            continue
        step = (int(histo.sum()) - int(histo[-1])) // 255
        if not step:
            lut[c] = np.arange(256)
            continue
        n = step // 2
        for i in range(256):
            lut[c, i] = min(n // step, 255)           # Image.point clips list entries to 8 bits
            n += int(h[i])
    return lut


def apply_op(img, op, magnitude, fill):
    """augment.py:10-87 for one uint8 RGB frame; fill = the RandAugment fill colour (3 ints)."""
    H, W, _ = img.shape
    if op == 'Identity':
        return img.copy()
    if op in ('ShearX', 'ShearY', 'TranslateX', 'TranslateY', 'Rotate'):
        m = op_matrix(op, magnitude, (W, H))
        if isinstance(m, str):
            return {'copy': img.copy(), 'rot180': img[::-1, ::-1].copy(),
                    'rot90': np.rot90(img, 1).copy(), 'rot270': np.rot90(img, 3).copy()}[m]
        return affine_bicubic(img, m, fill)
    if op == 'Brightness':
        return blend(np.zeros_like(img), img, 1.0 + magnitude)
    if op == 'Color':
        L = to_L(img)
        return blend(np.repeat(L[:, :, None], 3, axis=2), img, 1.0 + magnitude)
    if op == 'Contrast':
        L = to_L(img)
        mean = int(L.astype(np.float64).sum() / L.size + 0.5)     # ImageStat.Stat(...).mean[0] + 0.5
        return blend(np.full_like(img, mean), img, 1.0 + magnitude)
    if op == 'Sharpness':
        return blend(smooth(img), img, 1.0 + magnitude)
    if op == 'Posterize':
        mask = ~(2 ** (8 - int(magnitude)) - 1)
        lut = np.array([i & mask for i in range(256)], dtype=np.int64).astype(np.uint8)
        return lut_apply(img, np.stack([lut] * 3))
    if op == 'Solarize':
        lut = np.array([i if i < magnitude else 255 - i for i in range(256)], dtype=np.uint8)
        return lut_apply(img, np.stack([lut] * 3))
    if op == 'AutoContrast':
        return lut_apply(img, autocontrast_lut(img))
    if op == 'Equalize':
        return lut_apply(img, equalize_lut(img))
    raise ValueError(f'The provided operator {op} is not recognized.')


def randaugment(frames, ops, fill):
    """augment.py:159-193: the same op list applied to every frame of a sample."""
    out = []
    for f in frames:
        for name, mag in ops:
            f = apply_op(f, name, mag, fill)
        out.append(f)
    return np.stack(out)
