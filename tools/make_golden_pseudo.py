"""Golden vectors for the pseudo-label filter from the REFERENCE's own gen_data.py (build container only).

gen_data.main() is run end to end with in-memory stand-ins for everything around the selection logic
(`clip`, `nerv`, `models.build_model`, `datasets.build_dataset`; `.cuda()` is a no-op): a fake classifier
returns prescribed per-view probabilities, and the symlink tree main() writes (one folder per pseudo-label)
says, per sample, whether it was selected and with which label -- for every combination of --tta /
--tta_consistent / --tta_min_prob / --conf_thresh / --topk.  Writes tests/golden/pseudo_label.npz.

    python tools/make_golden_pseudo.py
"""
import argparse
import importlib.util
import os
import shutil
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

torch.Tensor.cuda = lambda self, *a, **k: self          # no GPU here; main() moves every batch tensor

STATE = {}


class FakeModel(torch.nn.Module):
    def forward(self, data_dict):
        return {'probs': data_dict['probs_in'].flatten(0, 1) if data_dict['probs_in'].dim() == 3
                else data_dict['probs_in']}

    def cuda(self):
        return self


class FakeEventDataset:
    new_cnames = None


def install_stubs():
    clip = types.ModuleType('clip')
    clip.load = lambda arch, device=None: (types.SimpleNamespace(visual=types.SimpleNamespace(output_dim=8)), None)
    sys.modules['clip'] = clip
    nerv = types.ModuleType('nerv')
    tr = types.ModuleType('nerv.training')

    class BaseDataModule:
        def __init__(self, params, train_set=None, val_set=None, use_ddp=False):
            self.val_loader = val_set.loader
    tr.BaseDataModule = BaseDataModule
    ut = types.ModuleType('nerv.utils')

    class AverageMeter:
        def __init__(self):
            self.s, self.n = 0., 0

        def update(self, v, n=1):
            self.s += v * n
            self.n += n

        @property
        def avg(self):
            return self.s / max(self.n, 1)
    ut.AverageMeter, ut.load_obj = AverageMeter, lambda p: {}
    sys.modules.update({'nerv': nerv, 'nerv.training': tr, 'nerv.utils': ut})
    models = types.ModuleType('models')
    models.build_model = lambda params: FakeModel()
    sys.modules['models'] = models
    datasets = types.ModuleType('datasets')
    datasets.build_dataset = lambda params, val_only=False, gen_data=False, tta=False: STATE['test_set']
    sys.modules['datasets'] = datasets


def run_reference(ref, probs, labels, K, tta, consistent, min_prob, thresh, topk, batch=7):
    """-> int64 [B]: pseudo-label of every sample, -1 when it was not selected."""
    B = labels.shape[0]
    names = [f'cls{k:02d}' for k in range(K)]
    tmp = tempfile.mkdtemp()
    try:
        root = os.path.join(tmp, 'data', 'training')
        files = [os.path.join(root, names[int(labels[i])], f'{names[int(labels[i])]}_{i:04d}.npy') for i in range(B)]
        ev = FakeEventDataset()
        ev.labels, ev.labeled_files, ev.root = labels.numpy(), files, root
        loader = []
        for i0 in range(0, B, batch):
            sl = slice(i0, min(B, i0 + batch))
            n = sl.stop - sl.start
            d = {'data_idx': torch.arange(sl.start, sl.stop), 'label': labels[sl],
                 'probs_in': probs[sl]}
            d['img'] = torch.zeros(n, 4, 1) if tta else torch.zeros(n, 1)
            d['valid_mask'] = torch.ones(n, 4, dtype=torch.bool) if tta else torch.ones(n, dtype=torch.bool)
            loader.append(d)
        STATE['test_set'] = types.SimpleNamespace(event_dataset=ev, classes=names, loader=loader)
        ref.args = argparse.Namespace(tta=tta, tta_consistent=consistent, tta_min_prob=min_prob,
                                      conf_thresh=thresh, topk=topk, weight='', gt_shots=0, params='fixture')
        ref.is_zs, ref.save_path = True, os.path.join(tmp, 'out')
        params = types.SimpleNamespace(clip_dict=dict(arch='ViT-B/32'), dataset='n_caltech', data_transforms=None)
        ref.main(params)
        out = -np.ones(B, dtype=np.int64)
        train = os.path.join(tmp, 'out', 'training')
        for k, name in enumerate(names):
            for fn in os.listdir(os.path.join(train, name)):
                out[int(fn.split('_')[-1].split('.')[0])] = k
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    install_stubs()
    spec = importlib.util.spec_from_file_location('ref_gen_data', '/root/reference/gen_data.py')
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    g = torch.Generator().manual_seed(123)
    B, K = 60, 6
    labels = torch.randint(0, K, (B,), generator=g)
    base = torch.randn(B, 1, K, generator=g) * 1.5
    base[torch.arange(B), 0, labels] += 1.0
    probs4 = (base + torch.randn(B, 4, K, generator=g) * 0.9).softmax(-1)        # [B, 4, K]
    out = dict(probs4=probs4.numpy(), labels=labels.numpy(), K=np.array(K))
    cases = []
    import contextlib
    import io
    for tta in (False, True):
        for consistent, min_prob in ((False, False), (True, False), (False, True), (True, True)):
            if not tta and (consistent or min_prob):
                continue
            for thresh in (-1.0, 0.4, 0.7):
                for topk in (0, 3):
                    probs = probs4 if tta else probs4[:, 0]
                    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
                        sel = run_reference(ref, probs, labels, K, tta, consistent, min_prob, thresh, topk)
                    tag = f'tta{int(tta)}_c{int(consistent)}_m{int(min_prob)}_t{thresh}_k{topk}'
                    out['sel_' + tag] = sel
                    cases.append(tag)
    out['cases'] = np.array(cases)
    path = os.path.join(ROOT, 'tests', 'golden', 'pseudo_label.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB;', len(cases), 'cases;',
          'selected per case:', [int((out['sel_' + c] >= 0).sum()) for c in cases])


if __name__ == '__main__':
    main()
