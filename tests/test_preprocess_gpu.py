"""HIP preprocess against the Pillow-pinned oracle: integer-exact (MI355X)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('shape,n_px', [((180, 240), 224), ((100, 120), 224), ((480, 640), 224),
                                        ((480, 640), 336), ((180, 240), 336), ((224, 224), 224),
                                        ((300, 224), 224)])
def test_resize_crop_bit_exact(shape, n_px, hip):
    import torch
    from eventclip_amd import preprocess as pp
    from oracle import preprocess as op
    rng = np.random.default_rng(shape[1] + n_px)
    frames = rng.integers(0, 256, size=(3, *shape, 3), dtype=np.uint8)
    got = pp.preprocess_frames(torch.from_numpy(frames).cuda(), n_px, mode='u8').cpu().numpy()
    np.testing.assert_array_equal(got, op.resize_crop_u8(frames, n_px))


def test_chw_tensor_equals_reference_transform(hip):
    import torch
    from eventclip_amd import preprocess as pp
    from eventclip_amd.synthetic import make_events
    from oracle import events as oe
    from oracle import preprocess as op
    frames = oe.events2frames(make_events(60000, (180, 240), 5), 'event_count', 'event_histogram',
                              shape=(180, 240), N=20000, grayscale=False)
    got = pp.preprocess_frames(torch.from_numpy(frames).cuda(), 224, mode='chw').cpu().numpy()
    want = op.preprocess(frames, 224)
    assert got.dtype == np.float32
    np.testing.assert_array_equal(got, want)      # same fp32 values, bit for bit


@pytest.mark.parametrize('dt', ['float16', 'bfloat16'])
@pytest.mark.parametrize('patch,n_px', [(14, 224), (32, 224), (16, 224), (14, 336)])
def test_patch_rows_equal_rounded_transform(patch, n_px, dt, hip):
    import torch
    from eventclip_amd import preprocess as pp
    from oracle import preprocess as op
    dtype = getattr(torch, dt)
    rng = np.random.default_rng(patch)
    frames = rng.integers(0, 256, size=(2, 100, 120, 3), dtype=np.uint8)
    k = 3 * patch * patch
    kpad = ((2 * k + 63) // 64) * 64              # [hi | lo | 0] rows
    out = torch.full((2, (n_px // patch) ** 2, kpad), 7., dtype=dtype, device='cuda')
    got = pp.preprocess_frames(torch.from_numpy(frames).cuda(), n_px, mode='patches', patch=patch,
                               kpad=kpad, dtype=dtype, out=out)
    ref = op.preprocess(frames, n_px)
    want = op.patchify_split(ref, patch, kpad, dtype)
    assert torch.equal(got.cpu(), want)           # incl. zeroed K padding
    # hi + lo carries the fp32 pixel value to 2^-22 (f16) / 2^-16 (bf16) relative
    full = torch.from_numpy(op.patchify(ref, patch, k))
    back = got[:, :, :k].float().cpu() + got[:, :, k:2 * k].float().cpu()
    assert float((back - full).abs().max()) < (2e-6 if dt == 'float16' else 1e-4)


def test_patchify_matches_oracle(hip):
    import ctypes
    import torch
    from eventclip_amd import _lib
    from oracle import preprocess as op
    x = torch.randn(3, 3, 224, 224, device='cuda')
    out = torch.empty(3, 256, 1216, dtype=torch.float16, device='cuda')
    _lib.check(_lib.lib().ec_patchify(_lib.ptr(x), 3, 224, 14, 1216, _lib.ptr(out), _lib.EC_F16,
                                      _lib.stream_ptr()))
    want = op.patchify_split(x.cpu().numpy(), 14, 1216, torch.float16)
    assert torch.equal(out.cpu(), want)


def test_preprocess_callable_is_drop_in(hip):
    from PIL import Image
    from eventclip_amd.preprocess import Preprocess
    from oracle import preprocess as op
    rng = np.random.default_rng(1)
    arr = rng.integers(0, 256, size=(180, 240, 3), dtype=np.uint8)
    t = Preprocess(224)(Image.fromarray(arr))
    assert tuple(t.shape) == (3, 224, 224)
    np.testing.assert_array_equal(t.cpu().numpy(), op.preprocess(arr[None], 224)[0])
