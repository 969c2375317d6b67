"""Host side of the device RandAugment: the sampler's draws and the descriptors handed to the kernel
(eventclip_amd/randaugment.py) against the reference-generated fixture and the Pillow-pinned oracle."""
import os

import numpy as np

from conftest import GOLDEN
from eventclip_amd import _lib
from eventclip_amd import randaugment as ra
from oracle import randaugment as ora


def test_sampler_draws_what_the_reference_draws():
    import torch
    z = np.load(os.path.join(GOLDEN, 'randaugment.npz'))
    for shape in ((180, 240), (480, 640)):
        names, mags = z[f'sample_{shape[0]}x{shape[1]}_names'], z[f'sample_{shape[0]}x{shape[1]}_mags']
        for seed in range(len(names)):
            aug = ra.RandAugment(num_ops=2, interpolation='bicubic', fill=None)
            torch.manual_seed(seed)
            aug.randomize_ops(shape)
            assert [o[0] for o in aug.cur_ops] == names[seed].tolist()
            assert [o[1] for o in aug.cur_ops] == mags[seed].tolist()
            state = torch.get_rng_state()
            torch.manual_seed(seed)
            ora.randomize_ops(shape)
            assert torch.equal(state, torch.get_rng_state())       # same number of draws consumed


def test_descriptors_carry_the_oracles_matrices_and_parameters():
    for (H, W) in ((180, 240), (100, 120), (64, 64)):
        for op in ra.OP_NAMES:
            table = ora.magnitude_table(op, (H, W))
            mags = [0.0] if table is None else [float(v) for v in table.tolist()]
            if op in ora.SIGNED:
                mags += [-m for m in mags if m]
            if op == 'Rotate':
                mags += [90.0, 180.0, 270.0]
            for mag in mags:
                d = ra.op_descriptor(op, mag, (H, W))
                if op in ('ShearX', 'ShearY', 'TranslateX', 'TranslateY', 'Rotate'):
                    m = ora.op_matrix(op, mag, (W, H))
                    if isinstance(m, str):
                        want = {'copy': _lib.EC_AUG_IDENTITY, 'rot180': _lib.EC_AUG_ROT180,
                                'rot90': _lib.EC_AUG_ROT90, 'rot270': _lib.EC_AUG_ROT270}[m]
                        assert d.kind == want
                    else:
                        assert d.kind == _lib.EC_AUG_AFFINE and list(d.m) == [float(v) for v in m]
                elif op in ('Brightness', 'Color', 'Contrast', 'Sharpness'):
                    assert d.alpha == np.float32(1.0 + mag)
                elif op == 'Posterize':
                    assert d.kind == _lib.EC_AUG_POSTERIZE and d.param == int(mag)
                elif op == 'Solarize':
                    assert d.kind == _lib.EC_AUG_SOLARIZE and d.param == mag


def test_struct_size():
    import ctypes
    assert ctypes.sizeof(_lib.EcAugOp) == 64
