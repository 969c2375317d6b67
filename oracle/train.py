"""Oracle for the few-shot `text-identity` training step (TEST INFRASTRUCTURE).

Restates, with explicit formulas in float64 (no autograd), what the reference computes for
adapter_type='text-identity' in train mode:
  forward   models/clip_cls.py:302-350 (identity adapter, F.normalize of the image features, invalid
            views zeroed, text_feats = F.normalize(parameter) :285-288, logits :332, aggregation
            :104-129) and calc_train_loss :164-175 (cross-entropy on the aggregated logits, or NLL of
            log(probs + 1e-6));
  backward  d loss / d text_feats (what loss.backward() leaves in text_feats.grad).
Pinned by tests/golden/train_text_identity.npz, produced by running the reference's own
FSCLIPClassifier under torch autograd (tools/make_golden_train.py).
adam_step / cosine_warmup_lr restate torch.optim.Adam (the `optimizer = 'Adam'` of the reference's
configs) and the warm-up + cosine schedule of method.py:82-98; the latter lives in the absent `nerv`
package: PARITY UNPINNED for the schedule.
"""
import math

import numpy as np


def _softmax(x):
    x = x - x.max(-1, keepdims=True)
    e = np.exp(x)
    return e / e.sum(-1, keepdims=True)


def fs_text_loss_and_grad(feats, valid, labels, text_param, logit_scale, agg='mean', probs_loss=False):
    """feats [B, T, D] raw image features (any value on invalid views), valid [B, T] bool,
    labels [B], text_param [K, D] -> (loss, grad [K, D], aggregated logits [B, K])."""
    f = np.asarray(feats, dtype=np.float64)
    m = np.asarray(valid, dtype=np.float64)
    t = np.asarray(text_param, dtype=np.float64)
    B, T, D = f.shape
    fn = f / np.maximum(np.linalg.norm(f, axis=-1, keepdims=True), 1e-12)   # :325-327
    fn = fn * m[..., None]                                                   # :329
    tn_norm = np.maximum(np.linalg.norm(t, axis=-1, keepdims=True), 1e-12)
    u = t / tn_norm                                                          # :287
    L = logit_scale * fn @ u.T                                               # :332  [B, T, K]
    n = m.sum(1, keepdims=True)
    if agg == 'sum':
        logits, w = L.sum(1), np.ones((B, T))
    elif agg == 'mean':
        logits, w = L.sum(1) / n, np.ones((B, T)) / n
    else:
        raise NotImplementedError(agg)          # 'max' raises upstream (clip_cls.py:117)
    onehot = np.eye(t.shape[0])[np.asarray(labels)]
    if not probs_loss:                                                       # :171
        p = _softmax(logits)
        loss = -np.log((p * onehot).sum(-1)).mean()
        dL = ((p - onehot) / B)[:, None, :] * w[..., None]
    else:                                                                    # :172-174
        pv = _softmax(L)                                                     # per view, :125
        P = (pv * m[..., None]).sum(1) / n
        Py = (P * onehot).sum(-1)
        loss = -np.log(Py + 1e-6).mean()
        dPy = -1.0 / (B * (Py + 1e-6))
        pvy = (pv * onehot[:, None, :]).sum(-1)                              # [B, T]
        dL = (m / n * dPy[:, None] * pvy)[..., None] * (onehot[:, None, :] - pv)
    dU = logit_scale * np.einsum('btk,btd->kd', dL, fn)
    grad = (dU - u * (u * dU).sum(-1, keepdims=True)) / tn_norm
    return float(loss), grad, logits


def adam_step(param, grad, exp_avg, exp_avg_sq, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.):
    """torch.optim.Adam (no amsgrad), float64 in place; step counts from 1."""
    if weight_decay:
        grad = grad + weight_decay * param
    exp_avg *= beta1
    exp_avg += (1 - beta1) * grad
    exp_avg_sq *= beta2
    exp_avg_sq += (1 - beta2) * grad * grad
    bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
    param -= lr / bc1 * exp_avg / (np.sqrt(exp_avg_sq) / math.sqrt(bc2) + eps)
    return param


def cosine_warmup_lr(step, total_steps, max_lr, min_lr, warmup_steps):
    """One cycle of linear warm-up then cosine decay (method.py:82-98: max_lr = lr, min_lr = lr / 100)."""
    if step < warmup_steps:
        return min_lr + (max_lr - min_lr) * step / max(warmup_steps, 1)
    frac = (step - warmup_steps) / max(total_steps - warmup_steps, 1)
    return min_lr + (max_lr - min_lr) * (1 + math.cos(math.pi * min(frac, 1.0))) / 2


def fs_trans_loss_and_grads(sd, feats, valid, labels, text_param, logit_scale, heads, residual, agg='mean',
                            probs_loss=False, dropout_p=0., masks=None):
    """'text-trans': loss and gradients of every adapter parameter and of text_feats, from torch
    float64 autograd over the oracle's own explicit forward (oracle/adapter.py's math, not the
    nn.Module).  sd: adapter state dict (reference names).  Pinned against the reference's
    FSCLIPClassifier under autograd by tests/golden/train_text_trans.npz.
    masks (with dropout_p > 0): {(layer, site): keep mask} for the four train-mode dropouts of
    nn.TransformerEncoderLayer, site 0 attention weights [B, heads, T, T], 1 after out_proj [B*T, d],
    2 inside the MLP [B*T, ffn], 3 after linear2 [B*T, d]; applied as x * keep / (1 - p)."""
    import torch
    import torch.nn.functional as F
    p = {k: torch.tensor(np.asarray(v), dtype=torch.float64, requires_grad=True) for k, v in sd.items()}
    t = torch.tensor(np.asarray(text_param), dtype=torch.float64, requires_grad=True)
    x0 = torch.tensor(np.asarray(feats), dtype=torch.float64)
    m = torch.tensor(np.asarray(valid), dtype=torch.bool)
    y = torch.tensor(np.asarray(labels), dtype=torch.long)
    B, T, _ = x0.shape
    x = F.linear(x0, p['in_proj.weight'], p['in_proj.bias'])
    dm = x.shape[-1]
    hd = dm // heads
    n_layers = len({k.split('.')[2] for k in sd if k.startswith('transformer_encoder.layers.')})
    key_mask = torch.zeros(B, 1, 1, T, dtype=torch.float64).masked_fill(~m[:, None, None, :], float('-inf'))

    def drop(v, layer, site):
        if not dropout_p:
            return v
        keep = torch.tensor(np.asarray(masks[(layer, site)]), dtype=torch.float64).reshape(v.shape)
        return v * keep / (1. - dropout_p)

    for i in range(n_layers):
        q = f'transformer_encoder.layers.{i}.'
        h = F.layer_norm(x, (dm,), p[q + 'norm1.weight'], p[q + 'norm1.bias'], 1e-5)
        qkv = F.linear(h, p[q + 'self_attn.in_proj_weight'], p[q + 'self_attn.in_proj_bias'])
        qq, kk, vv = qkv.split(dm, dim=-1)
        qq = qq.view(B, T, heads, hd).transpose(1, 2) * hd ** -0.5
        kk = kk.view(B, T, heads, hd).transpose(1, 2)
        vv = vv.view(B, T, heads, hd).transpose(1, 2)
        att = drop((qq @ kk.transpose(-1, -2) + key_mask).softmax(-1), i, 0)
        o = (att @ vv).transpose(1, 2).reshape(B, T, dm)
        sa = F.linear(o, p[q + 'self_attn.out_proj.weight'], p[q + 'self_attn.out_proj.bias'])
        x = x + drop(sa.reshape(B * T, dm), i, 1).reshape(B, T, dm)
        h = F.layer_norm(x, (dm,), p[q + 'norm2.weight'], p[q + 'norm2.bias'], 1e-5)
        h = F.relu(F.linear(h, p[q + 'linear1.weight'], p[q + 'linear1.bias']))
        h = drop(h.reshape(B * T, -1), i, 2).reshape(B, T, -1)
        ff = F.linear(h, p[q + 'linear2.weight'], p[q + 'linear2.bias'])
        x = x + drop(ff.reshape(B * T, dm), i, 3).reshape(B, T, dm)
    new = F.linear(x, p['out_proj.weight'], p['out_proj.bias'])
    mixed = x0 * residual + new * (1. - residual)
    fn = F.normalize(mixed, p=2, dim=-1) * m[..., None]
    L = logit_scale * fn @ F.normalize(t, p=2, dim=-1).T
    n = m.double().sum(1, keepdim=True)
    logits = L.sum(1) if agg == 'sum' else L.sum(1) / n
    if not probs_loss:
        loss = F.cross_entropy(logits, y)
    else:
        probs = (L.softmax(-1) * m[..., None]).sum(1) / n
        loss = F.nll_loss((probs + 1e-6).log(), y)
    loss.backward()
    grads = {k: v.grad.numpy() for k, v in p.items()}
    grads['text_feats'] = t.grad.numpy()
    return float(loss.detach()), grads, logits.detach().numpy()
