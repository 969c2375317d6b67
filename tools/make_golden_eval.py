"""Golden accuracies from the REFERENCE's own test.py main() (build container only): the evaluation loop is
run end to end with the stand-ins of tools/make_golden_pseudo.py around it (a fake classifier returns
prescribed logits / probs for uneven batches; nerv's AverageMeter is the sum(acc * n) / sum(n) stand-in),
for N-Caltech-style (top-1) and N-ImageNet-style (top-1 and top-5) runs.  Writes tests/golden/eval_meters.npz.

    python tools/make_golden_eval.py
"""
import argparse
import contextlib
import importlib.util
import io
import os
import re
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden_pseudo as mp   # noqa: E402  (stubs; also makes .cuda() a no-op)

ROOT = mp.ROOT


class FakeClassifier(torch.nn.Module):
    def forward(self, d):
        return {'probs': d['probs_in'], 'logits': d['logits_in']}

    def cuda(self):
        return self


def main():
    mp.install_stubs()
    sys.modules['models'].build_model = lambda params: FakeClassifier()
    sys.modules['datasets'].build_dataset = lambda params, val_only=False, subset=None: mp.STATE['test_set']
    spec = importlib.util.spec_from_file_location('ref_test', '/root/reference/test.py')
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    g = torch.Generator().manual_seed(5)
    out = {}
    for name, dataset, K, sizes in (('ncaltech', 'n_caltech', 11, (7, 7, 3)), ('nin', 'n_imagenet', 40, (16, 16, 16, 5))):
        B = sum(sizes)
        labels = torch.randint(0, K, (B,), generator=g)
        logits = torch.randn(B, K, generator=g) * 2
        logits[torch.arange(B), labels] += 1.5
        probs = (logits * 0.7 + torch.randn(B, K, generator=g)).softmax(-1)     # a different ranking than logits
        loader, i0 = [], 0
        for n in sizes:
            loader.append({'label': labels[i0:i0 + n], 'probs_in': probs[i0:i0 + n], 'logits_in': logits[i0:i0 + n]})
            i0 += n
        mp.STATE['test_set'] = types.SimpleNamespace(classes=[str(k) for k in range(K)], loader=loader)
        ref.args = argparse.Namespace(subset=-1, weight='', params='fixture')
        ref.is_zs = True
        params = types.SimpleNamespace(clip_dict=dict(arch='ViT-B/32'), dataset=dataset, data_transforms=None)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
            ref.main(params)
        vals = [float(v) for v in re.findall(r'accuracy@\d: ([\d.]+)%', buf.getvalue())]
        p1, l1 = ref.main(params, printing=False) if False else (None, None)
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            p1, l1 = ref.main(params, printing=False)
        out[name + '_labels'], out[name + '_logits'], out[name + '_probs'] = labels.numpy(), logits.numpy(), probs.numpy()
        out[name + '_sizes'] = np.array(sizes)
        out[name + '_acc1'] = np.array([p1, l1])                 # exact (probs, logits)
        out[name + '_printed'] = np.array(vals)                  # percentages as printed (2 decimals)
        print(name, 'acc@1 (probs, logits):', p1, l1, 'printed:', vals)
    path = os.path.join(ROOT, 'tests', 'golden', 'eval_meters.npz')
    np.savez_compressed(path, **out)
    print('wrote', path)


if __name__ == '__main__':
    main()
