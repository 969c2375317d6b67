"""Golden vectors for the few-shot `text-identity` training step: the
REFERENCE's own FSCLIPClassifier (models/clip_cls.py, imported with the stubs of
tools/make_golden_models.py) in train mode: loss from calc_train_loss and d loss / d text_feats
from torch autograd, for both losses and the aggregation functions that run upstream.
Writes tests/golden/train_text_identity.npz.

    python tools/make_golden_train.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden_models as mg   # noqa: E402  (installs the stubs, loads the reference modules)

GOLD = mg.GOLD


def main():
    torch.manual_seed(7)
    C, K, R = 48, 9, 4
    table = torch.nn.functional.normalize(torch.randn(100, C), dim=-1) * 3
    names = [f'class_{i}' for i in range(K)]
    out = dict(C=np.array(C), K=np.array(K))
    cases = []
    for ci, (B, T) in enumerate(((6, 5), (3, 1), (8, 10))):
        valid = torch.rand(B, T) < 0.6
        valid[:, 0] = True
        imgs = torch.randn(B, T, 3, R, R) * valid[:, :, None, None, None]
        labels = torch.randint(0, K, (B,))
        for agg in ('sum', 'mean'):
            for probs_loss in (False, True):
                torch.manual_seed(11)
                model = mg.ref_cls.FSCLIPClassifier(
                    adapter_dict=dict(adapter_type='text-identity', in_dim=C, residual=True),
                    clip_dict=dict(clip_model=mg.FakeCLIP(C, table.clone()),
                                   prompt='a point cloud image of a {}', class_names=names, agg_func=agg),
                    loss_dict=dict(use_logits_loss=not probs_loss, use_probs_loss=probs_loss)).train()
                with torch.no_grad():
                    model.text_feats.add_(torch.randn_like(model.text_feats) * 0.3)
                assert model.text_feats.requires_grad
                data = {'img': imgs, 'valid_mask': valid, 'label': labels}
                o = model(data)
                loss = model.calc_train_loss(data, o)['ce_loss']
                loss.backward()
                tag = f'c{ci}_{agg}_{"probs" if probs_loss else "logits"}'
                out[tag + '_loss'] = loss.detach().numpy()
                out[tag + '_grad'] = model.text_feats.grad.numpy().copy()
                out[tag + '_logits'] = o['logits'].detach().numpy()
                if agg == 'sum' and not probs_loss:
                    # the image features the classifier saw (what a feature cache would hold)
                    with torch.no_grad():
                        feats = torch.zeros(B, T, C)
                        feats[valid] = model.get_img_feats(imgs[valid])
                    out[f'c{ci}_feats'] = feats.numpy()
                    out[f'c{ci}_valid'] = valid.numpy()
                    out[f'c{ci}_labels'] = labels.numpy()
                    out[f'c{ci}_text_param'] = model.text_feats.detach().numpy().copy()
                    out[f'c{ci}_logit_scale'] = np.array(float(model.logit_scale))
        cases.append((B, T))
    out['cases'] = np.array(cases)
    path = os.path.join(GOLD, 'train_text_identity.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB')
    trans_fixture(table, names, C, K, R)


def trans_fixture(table, names, C, K, R):
    """adapter_type='text-trans': gradients of every adapter parameter and of text_feats.  The
    classifier is put in eval mode (dropout of nn.TransformerEncoderLayer off: its random masks cannot
    be reproduced) but keeps requires_grad, so autograd differentiates the deterministic function."""
    out = dict(C=np.array(C), K=np.array(K))
    ad_cfg = dict(adapter_type='text-trans', in_dim=C, d_model=32, num_heads=2, ffn_dim=64,
                  norm_first=True, num_layers=2, residual=0.8)
    cases = []
    for ci, (B, T) in enumerate(((5, 4), (4, 1), (6, 10))):
        torch.manual_seed(40 + ci)
        valid = torch.rand(B, T) < 0.6
        valid[:, 0] = True
        imgs = torch.randn(B, T, 3, R, R) * valid[:, :, None, None, None]
        labels = torch.randint(0, K, (B,))
        for agg, probs_loss in (('sum', False), ('mean', True)):
            torch.manual_seed(13)
            model = mg.ref_cls.FSCLIPClassifier(
                adapter_dict=dict(ad_cfg),
                clip_dict=dict(clip_model=mg.FakeCLIP(C, table.clone()),
                               prompt='a point cloud image of a {}', class_names=names, agg_func=agg),
                loss_dict=dict(use_logits_loss=not probs_loss, use_probs_loss=probs_loss)).eval()
            with torch.no_grad():
                model.text_feats.add_(torch.randn_like(model.text_feats) * 0.3)
                for p in model.adapter.parameters():
                    p.add_(torch.randn_like(p) * 0.05)
            data = {'img': imgs, 'valid_mask': valid, 'label': labels}
            o = model(data)
            loss = model.calc_train_loss(data, o)['ce_loss']
            loss.backward()
            tag = f'c{ci}_{agg}_{"probs" if probs_loss else "logits"}'
            out[tag + '_loss'] = loss.detach().numpy()
            out[tag + '_logits'] = o['logits'].detach().numpy()
            out[tag + '_g:text_feats'] = model.text_feats.grad.numpy().copy()
            for k, p in model.adapter.named_parameters():
                out[tag + '_g:' + k] = p.grad.numpy().copy()
            if not probs_loss:
                with torch.no_grad():
                    feats = torch.zeros(B, T, C)
                    feats[valid] = model.get_img_feats(imgs[valid])
                out[f'c{ci}_feats'], out[f'c{ci}_valid'] = feats.numpy(), valid.numpy()
                out[f'c{ci}_labels'] = labels.numpy()
                out[f'c{ci}_text_param'] = model.text_feats.detach().numpy().copy()
                out[f'c{ci}_logit_scale'] = np.array(float(model.logit_scale))
                if ci == 0:
                    for k, p in model.adapter.state_dict().items():
                        out['w:' + k] = p.numpy().copy()
        cases.append((B, T))
    out['cases'] = np.array(cases)
    for k, v in ad_cfg.items():
        out['adcfg_' + k] = np.array(v)
    path = os.path.join(GOLD, 'train_text_trans.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()
