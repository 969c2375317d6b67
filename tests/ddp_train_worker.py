"""Worker of tests/test_train_gpu.py::test_two_rank_training_equals_one_rank_full_batch (not a test itself)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(seed=0):
    from eventclip_amd.adapter import TransformerAdapter
    torch.manual_seed(seed)
    ad = TransformerAdapter(in_dim=64, d_model=32, num_heads=2, ffn_dim=64, num_layers=2, residual=0.5)
    with torch.no_grad():
        for p in ad.parameters():
            p.add_(torch.randn_like(p) * 0.05)
    clf = type('Clf', (), {})()
    clf.adapter, clf.adapter_type, clf.prompt_tuning = ad.cuda(), 'trans', True
    clf.text_feats = torch.nn.Parameter(torch.randn(5, 64).cuda())
    clf.logit_scale, clf.agg_func, clf.use_probs_loss = 100.0, 'mean', False
    g = torch.Generator().manual_seed(1)
    B, T = 8, 3
    valid = torch.rand(B, T, generator=g) < 0.8
    valid[:, 0] = True
    labels = torch.randint(0, 5, (B,), generator=g)
    feats = torch.randn(B, T, 64, generator=g) * valid[..., None]
    return clf, feats.cuda(), valid.cuda(), labels.cuda()


def main():
    from eventclip_amd.train import AdapterTrainer
    out = sys.argv[1]
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    torch.cuda.set_device(0)
    if world > 1:
        dist.init_process_group('gloo')
    clf, feats, valid, labels = build()
    sl = slice(rank * 4, rank * 4 + 4) if world > 1 else slice(0, 8)
    # a large Adam eps keeps elements whose true gradient is ~0 from turning fp noise into +-lr steps
    tr = AdapterTrainer(clf, lr=1e-2, total_steps=10, dropout=0., eps=1e-2)
    for _ in range(3):
        tr.step(feats[sl], valid[sl], labels[sl])
    if rank == 0:
        sd = {k: v.cpu().numpy() for k, v in clf.adapter.state_dict().items()}
        sd['text_feats'] = clf.text_feats.data.cpu().numpy()
        np.savez(out, **sd)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
