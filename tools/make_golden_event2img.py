"""Golden vectors from the REFERENCE's own datasets/event2img.py Event2ImageDataset (build container only):
view counts, zero padding, valid masks, the random subset when a sample has more chunks than max_imgs, and
the order / content of the four TTA views, with an identity `transforms` (uint8 frame -> CHW tensor) so the
fixture isolates the dataset wrapper from CLIP's preprocess.  Writes tests/golden/event2img.npz.

    python tools/make_golden_event2img.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from eventclip_amd.synthetic import make_events  # noqa: E402

pkg = types.ModuleType('datasets')
pkg.__path__ = ['/root/reference/datasets']
sys.modules['datasets'] = pkg
aug = types.ModuleType('datasets.augment')          # torchvision-based RandAugment: not exercised (augment=False)
aug.RandAugment, aug.InterpolationMode = object, types.SimpleNamespace(BICUBIC=3)
sys.modules['datasets.augment'] = aug


def load(name):
    spec = importlib.util.spec_from_file_location('datasets.' + name, f'/root/reference/datasets/{name}.py')
    m = importlib.util.module_from_spec(spec)
    sys.modules['datasets.' + name] = m
    spec.loader.exec_module(m)
    return m


load('vis')
load('utils')
e2i = load('event2img')


class FakeEvents:
    classes = ['a', 'b']
    resolution = (36, 52)
    max_t, max_n = 0.3, 2600            # round(2600 / 600) = 4 views
    augmentation, num_shots, root = False, None, '/data/train'

    def __init__(self, samples):
        self.samples = samples

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, i):
        return {'events': self.samples[i].copy(), 'label': i % 2}


def to_tensor(img):
    return torch.from_numpy(np.asarray(img).copy()).permute(2, 0, 1)          # uint8 CHW


def main():
    res = FakeEvents.resolution
    counts = [250, 600, 900, 901, 2400, 3100, 4000]      # < N, exactly N, remainder dropped / kept, full, more than max_imgs
    samples = [make_events(n, res, seed=70 + i, hot_pixels=1) for i, n in enumerate(counts)]
    qa = dict(max_imgs=10, split_method='event_count', convert_method='event_histogram', N=600, grayscale=True,
              count_non_zero=False, background_mask=True)
    out = {'counts': np.array(counts), 'resolution': np.array(res), 'max_n': np.array(FakeEvents.max_n)}
    for k, v in qa.items():
        out['qa_' + k] = np.array(v)
    for tta in (False, True):
        ds = e2i.Event2ImageDataset(to_tensor, FakeEvents(samples), quantize_args=dict(qa), tta=tta)
        out['max_imgs'] = np.array(ds.max_imgs)
        for i in range(len(samples)):
            torch.manual_seed(1000 + i)                   # _subsample_imgs draws torch.randperm when F > max_imgs
            d = ds[i]
            out[f'tta{int(tta)}_img{i}'] = d['img'].numpy()
            out[f'tta{int(tta)}_valid{i}'] = d['valid_mask'].numpy()
    for i, s in enumerate(samples):
        out[f'events{i}'] = s
    path = os.path.join(ROOT, 'tests', 'golden', 'event2img.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB; max_imgs', int(out['max_imgs']),
          'valid views:', [int(out[f'tta0_valid{i}'].sum()) for i in range(len(samples))])


if __name__ == '__main__':
    main()
