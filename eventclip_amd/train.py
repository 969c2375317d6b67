"""Few-shot training on cached image features, `text-identity` adapters.

The reference tunes `text_feats` (and, for `text-trans`, the transformer adapter) with the CLIP
encoder frozen (train.py, method.py, models/clip_cls.py:164-175, 281-350); with the encoder's
outputs cached per epoch one optimisation step is a few small fp32 kernels:
``fs_text_loss_grad`` (forward + loss + d loss / d text_feats in closed form, ``ec_fs_text_loss_grad``)
and ``adam_step`` (torch.optim.Adam, ``ec_adam_step``).  ``TextFeatTrainer`` strings them together with
the warm-up + cosine schedule of method.py:82-98 and, across ranks, an all-reduce of the K x D
gradient (the only trainable tensor of this adapter type).
"""
import math

import torch
import torch.distributed as dist

from . import _lib

_AGG = {'sum': _lib.EC_AGG_SUM, 'mean': _lib.EC_AGG_MEAN}
_WS = {}


def fs_text_loss_grad(img_feats, valid, labels, text_param, logit_scale, agg='sum', use_probs_loss=False,
                      return_logits=False):
    """img_feats fp32 CUDA [B, T, D] (raw encoder outputs), valid bool [B, T], labels int [B],
    text_param fp32 [K, D] (raw parameter).  Returns (loss 0-dim tensor, grad [K, D][, logits [B, K]])."""
    dev = _lib.require_gpu()
    if agg not in _AGG:
        raise NotImplementedError(f'agg_func {agg!r}: the reference trains with sum / mean')
    f = img_feats.float().contiguous()
    B, T, D = f.shape
    t = text_param.detach().float().contiguous()
    K = t.shape[0]
    assert t.shape[1] == D and valid.shape == (B, T) and labels.shape == (B,)
    v8 = valid.to(torch.uint8).contiguous()
    lab = labels.to(torch.int32).contiguous()
    need = int(_lib.lib().ec_fs_text_train_workspace_bytes(B, T, D, K))
    ws = _WS.get(dev.index)
    if ws is None or ws.numel() < need:
        ws = _WS[dev.index] = torch.empty((need,), dtype=torch.uint8, device=dev)
    loss = torch.empty((1,), dtype=torch.float32, device=dev)
    grad = torch.empty((K, D), dtype=torch.float32, device=dev)
    logits = torch.empty((B, K), dtype=torch.float32, device=dev) if return_logits else None
    rc = _lib.lib().ec_fs_text_loss_grad(_lib.ptr(f), _lib.ptr(v8), _lib.ptr(lab), _lib.ptr(t), B, T, D, K,
                                         float(logit_scale), _AGG[agg], int(bool(use_probs_loss)),
                                         _lib.ptr(loss), _lib.ptr(grad), _lib.ptr(logits), _lib.ptr(ws),
                                         ws.numel(), _lib.stream_ptr())
    _lib.check(rc, 'ec_fs_text_loss_grad')
    return (loss[0], grad, logits) if return_logits else (loss[0], grad)


_LAYER_FIELDS = (('ln1_g', 'norm1.weight'), ('ln1_b', 'norm1.bias'), ('qkv_w', 'self_attn.in_proj_weight'),
                 ('qkv_b', 'self_attn.in_proj_bias'), ('o_w', 'self_attn.out_proj.weight'),
                 ('o_b', 'self_attn.out_proj.bias'), ('ln2_g', 'norm2.weight'), ('ln2_b', 'norm2.bias'),
                 ('w1', 'linear1.weight'), ('b1', 'linear1.bias'), ('w2', 'linear2.weight'), ('b2', 'linear2.bias'))


def _adapter_struct(adapter, tensors):
    """ec_adapter_train_params over `tensors` (name -> fp32 CUDA tensor, the adapter's state-dict names)."""
    import ctypes
    layers = (_lib.EcAdapterTrainLayer * adapter.num_layers)()
    for i in range(adapter.num_layers):
        for field, name in _LAYER_FIELDS:
            setattr(layers[i], field, tensors[f'transformer_encoder.layers.{i}.{name}'].data_ptr())
    p = _lib.EcAdapterTrainParams()
    p.in_dim, p.d_model, p.heads = adapter.in_dim, adapter.d_model, adapter.num_heads
    p.ffn_dim, p.layers, p.residual = adapter.ffn_dim, adapter.num_layers, float(adapter.residual)
    p.in_w, p.in_b = tensors['in_proj.weight'].data_ptr(), tensors['in_proj.bias'].data_ptr()
    p.out_w, p.out_b = tensors['out_proj.weight'].data_ptr(), tensors['out_proj.bias'].data_ptr()
    p.blocks = ctypes.cast(layers, ctypes.POINTER(_lib.EcAdapterTrainLayer))
    return p, layers


def dropout_mask(seed, site, n, p):
    """Keep mask (uint8 CUDA [n]) of dropout site ``site`` = 4 * layer + {0 attention weights, 1 after
    out_proj, 2 inside the MLP, 3 after linear2} as ``fs_trans_loss_grad`` draws it."""
    dev = _lib.require_gpu()
    m = torch.empty((int(n),), dtype=torch.uint8, device=dev)
    _lib.check(_lib.lib().ec_dropout_mask(int(seed), int(site), int(n), float(p), _lib.ptr(m), _lib.stream_ptr()),
               'ec_dropout_mask')
    return m


def fs_trans_loss_grad(img_feats, valid, labels, text_param, logit_scale, adapter, agg='sum',
                       use_probs_loss=False, return_logits=False, dropout_p=0., seed=0, out=None):
    """The `text-trans` step: img_feats fp32 CUDA [B, T, D] with ZERO rows on invalid views
    (clip_cls.py:319-321), ``adapter`` an eventclip_amd.adapter.TransformerAdapter on the GPU.
    Returns (loss, grads) with grads = {adapter state-dict name: tensor, 'text_feats': [K, D]}
    dropout_p > 0: the train-mode dropouts of nn.TransformerEncoderLayer (p = 0.1 upstream), masks from a
    stateless hash of (seed, site, element); 0: the deterministic eval-mode function.
    out (optional): {name: tensor} to receive the gradients (every adapter name and 'text_feats'; a trainer's
    fixed buffers -- nothing is allocated for them then)."""
    import ctypes
    dev = _lib.require_gpu()
    if agg not in _AGG:
        raise NotImplementedError(f'agg_func {agg!r}: the reference trains with sum / mean')
    f = img_feats.float().contiguous()
    B, T, D = f.shape
    t = text_param.detach().float().contiguous()
    K = t.shape[0]
    params = {k: v.detach() for k, v in adapter.named_parameters()}
    for k, v in params.items():
        assert v.is_cuda and v.dtype == torch.float32 and v.is_contiguous(), k
    grads = {k: out[k] for k in params} if out is not None else {k: torch.empty_like(v) for k, v in params.items()}
    ps, keep_p = _adapter_struct(adapter, params)
    gs, keep_g = _adapter_struct(adapter, grads)
    need = int(_lib.lib().ec_fs_trans_train_workspace_bytes(B, T, D, K, adapter.d_model, adapter.ffn_dim,
                                                            adapter.num_heads, adapter.num_layers))
    ws = _WS.get(dev.index)
    if ws is None or ws.numel() < need:
        ws = _WS[dev.index] = torch.empty((need,), dtype=torch.uint8, device=dev)
    loss = torch.empty((1,), dtype=torch.float32, device=dev)
    gtext = out['text_feats'] if out is not None and 'text_feats' in out else torch.empty((K, D), dtype=torch.float32, device=dev)
    assert gtext.is_contiguous() and tuple(gtext.shape) == (K, D) and gtext.dtype == torch.float32
    logits = torch.empty((B, K), dtype=torch.float32, device=dev) if return_logits else None
    v8 = valid.to(torch.uint8).contiguous()       # named: a temporary's block would be handed to the next one
    lab = labels.to(torch.int32).contiguous()
    rc = _lib.lib().ec_fs_trans_loss_grad(
        _lib.ptr(f), _lib.ptr(v8), _lib.ptr(lab),
        _lib.ptr(t), B, T, D, K, float(logit_scale), _AGG[agg], int(bool(use_probs_loss)), ctypes.byref(ps),
        ctypes.byref(gs), float(dropout_p), int(seed), _lib.ptr(loss), _lib.ptr(gtext), _lib.ptr(logits), _lib.ptr(ws), ws.numel(),
        _lib.stream_ptr())
    _lib.check(rc, 'ec_fs_trans_loss_grad')
    grads['text_feats'] = gtext
    return (loss[0], grads, logits) if return_logits else (loss[0], grads)


def adam_step(param, grad, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.):
    """In-place torch.optim.Adam update of a contiguous fp32 CUDA tensor."""
    _lib.require_gpu()
    for x in (param, grad, exp_avg, exp_avg_sq):
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.numel() == param.numel()
    rc = _lib.lib().ec_adam_step(_lib.ptr(param), _lib.ptr(grad), _lib.ptr(exp_avg), _lib.ptr(exp_avg_sq),
                                 param.numel(), float(lr), float(betas[0]), float(betas[1]), float(eps),
                                 float(weight_decay), int(step), _lib.stream_ptr())
    _lib.check(rc, 'ec_adam_step')


def _device_table(items):
    """ctypes array of structs -> device copy (uint8 CUDA tensor) for the batched kernels' item tables."""
    import numpy as np
    return torch.from_numpy(np.frombuffer(items, dtype=np.uint8).copy()).cuda()


def cosine_warmup_lr(step, total_steps, max_lr, min_lr, warmup_steps):
    """One cycle of linear warm-up then cosine decay (method.py:82-98: min_lr = lr / 100)."""
    if step < warmup_steps:
        return min_lr + (max_lr - min_lr) * step / max(warmup_steps, 1)
    frac = (step - warmup_steps) / max(total_steps - warmup_steps, 1)
    return min_lr + (max_lr - min_lr) * (1 + math.cos(math.pi * min(frac, 1.0))) / 2


class TextFeatTrainer:
    """Trains ``classifier.text_feats`` (FSCLIPClassifier built with adapter_type='text-identity')."""

    def __init__(self, classifier, lr, total_steps, warmup_steps_pct=0.05, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=0.):
        if getattr(classifier, 'adapter_type', None) != 'identity' or not getattr(classifier, 'prompt_tuning', False):
            raise NotImplementedError("TextFeatTrainer handles adapter_type='text-identity'")
        self.clf = classifier
        self.lr, self.total_steps = float(lr), int(total_steps)
        self.warmup_steps = warmup_steps_pct * self.total_steps          # method.py:86
        self.betas, self.eps, self.weight_decay = betas, float(eps), float(weight_decay)
        p = classifier.text_feats.data
        self.exp_avg, self.exp_avg_sq = torch.zeros_like(p), torch.zeros_like(p)
        self.steps = 0

    @torch.no_grad()
    def step(self, img_feats, valid, labels):
        """img_feats [B, T, D]: cached ``get_img_feats`` outputs scattered to [B, T] (zeros for padding)."""
        clf = self.clf
        loss, grad = fs_text_loss_grad(img_feats, valid, labels, clf.text_feats.data, clf.logit_scale,
                                       clf.agg_func, clf.use_probs_loss)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(grad)                                        # DDP: mean of the rank gradients
            grad /= dist.get_world_size()
        lr = cosine_warmup_lr(self.steps, self.total_steps, self.lr, self.lr / 100., self.warmup_steps)
        self.steps += 1
        adam_step(clf.text_feats.data, grad, self.exp_avg, self.exp_avg_sq, self.steps, lr, self.betas,
                  self.eps, self.weight_decay)
        if hasattr(clf, '_invalidate_text_cache'):
            clf._invalidate_text_cache()
        return loss


class AdapterTrainer:
    """Trains the TransformerAdapter (+ ``text_feats`` for 'text-trans') of an FSCLIPClassifier on
    cached encoder features: ``fs_trans_loss_grad`` + one ``adam_step`` per tensor, with the
    encoder layers' dropout (``dropout``, 0.1 like nn.TransformerEncoderLayer's default) reseeded per step."""

    def __init__(self, classifier, lr, total_steps, warmup_steps_pct=0.05, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=0., dropout=0.1, seed=0):
        if getattr(classifier, 'adapter_type', None) != 'trans':
            raise NotImplementedError("AdapterTrainer handles adapter_type='trans' / 'text-trans'")
        self.clf = classifier
        self.lr, self.total_steps = float(lr), int(total_steps)
        self.warmup_steps = warmup_steps_pct * self.total_steps
        self.betas, self.eps, self.weight_decay = betas, float(eps), float(weight_decay)
        self.dropout, self.seed = float(dropout), int(seed)
        self.tensors = {k: p.data for k, p in classifier.adapter.named_parameters()}
        if classifier.prompt_tuning:
            self.tensors['text_feats'] = classifier.text_feats.data
        self.state = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in self.tensors.items()}
        self.steps = 0
        # gradients at fixed addresses (views of one flat buffer: one all-reduce) and one Adam launch over all of them
        total = sum(v.numel() for v in self.tensors.values())
        self._flat = torch.zeros((total,), dtype=torch.float32, device=next(iter(self.tensors.values())).device)
        self._grads, off = {}, 0
        for k, v in self.tensors.items():
            assert v.is_contiguous()
            self._grads[k] = self._flat[off:off + v.numel()].view(v.shape)
            off += v.numel()
        if not classifier.prompt_tuning:
            self._grads['text_feats'] = torch.empty_like(classifier.get_text_feats().float())   # computed, not trained
        items = (_lib.EcAdamItem * len(self.tensors))()
        for it, (k, p) in zip(items, self.tensors.items()):
            m, v = self.state[k]
            it.param, it.grad, it.exp_avg, it.exp_avg_sq = p.data_ptr(), self._grads[k].data_ptr(), m.data_ptr(), v.data_ptr()
            it.n, it.group = p.numel(), 0
        self._items = _device_table(items)
        self._max_n = max(p.numel() for p in self.tensors.values())

    @torch.no_grad()
    def step(self, img_feats, valid, labels):
        clf = self.clf
        text = clf.text_feats.data if clf.prompt_tuning else clf.get_text_feats().float()
        loss, _ = fs_trans_loss_grad(img_feats, valid, labels, text, clf.logit_scale, clf.adapter,
                                     clf.agg_func, clf.use_probs_loss, dropout_p=self.dropout,
                                     seed=self.seed * 1000003 + self.steps, out=self._grads)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self._flat)
            self._flat /= dist.get_world_size()
        lr = cosine_warmup_lr(self.steps, self.total_steps, self.lr, self.lr / 100., self.warmup_steps)
        self.steps += 1
        rc = _lib.lib().ec_adam_step_multi(_lib.ptr(self._items), len(self.tensors), self._max_n, float(lr), float(lr),
                                           float(self.betas[0]), float(self.betas[1]), self.eps, self.weight_decay,
                                           int(self.steps), None, None, _lib.stream_ptr())
        _lib.check(rc, 'ec_adam_step_multi')
        clf.adapter._packed = None          # the forward kernel's transposed copies are stale now
        return loss
