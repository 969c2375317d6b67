"""Feature adapters with the reference's API (models/adapter.py), forward on HIP.

``TransformerAdapter`` keeps its parameters in the same torch modules as the
reference (``in_proj``, ``transformer_encoder.layers.{i}.*``, ``out_proj``) so that
checkpoints written by the reference's trainer load key for key; those modules are
parameter holders only -- ``forward`` runs ``ec_adapter_forward``.  fp32, eval only.
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib


class Adapter(nn.Module):
    """Base adapter: residual weight handling (adapter.py:5-32)."""

    def __init__(self, residual=True):
        super().__init__()
        assert isinstance(residual, (bool, float))
        if isinstance(residual, bool):
            residual = 0.5 if residual else 0.
        if isinstance(residual, float):
            assert 0. <= residual <= 1.
        self.residual = residual

    def residual_add(self, in_feats, new_feats):
        assert isinstance(self.residual, float)
        return in_feats * self.residual + new_feats * (1. - self.residual)

    def forward(self, *args, **kwargs):
        raise NotImplementedError

    @property
    def dtype(self):
        raise NotImplementedError

    # ---- fused entry used by FSCLIPClassifier.forward ----
    def forward_rows(self, feats, row_idx):
        """feats [Nv, C] (valid views, compact) + row_idx [B, T] -> [B, T, C]: the zero
        scatter of clip_cls.py:319-321 followed by ``forward``."""
        raise NotImplementedError


def _scatter(feats, row_idx):
    B, T = row_idx.shape
    full = torch.zeros((B, T, feats.shape[-1]), dtype=feats.dtype, device=feats.device)
    valid = row_idx >= 0
    full[valid] = feats[row_idx[valid].long()]
    return full, valid


class IdentityAdapter(Adapter):
    """Trivial adapter that does nothing (adapter.py:35-50)."""

    def __init__(self, *args, **kwargs):
        super().__init__(residual=False)
        self.dummy = nn.Parameter(torch.zeros(1), requires_grad=False)

    def forward(self, feats, valid_masks):
        return feats

    def forward_rows(self, feats, row_idx):
        return _scatter(feats, row_idx)[0]

    @property
    def dtype(self):
        return self.dummy.dtype


class TransformerAdapter(Adapter):
    """Order-invariant Transformer over the views of one sample (adapter.py:53-109)."""

    def __init__(self, in_dim, d_model=256, num_heads=4, ffn_dim=256 * 4, norm_first=True,
                 num_layers=2, residual=False):
        super().__init__(residual=residual)
        if not norm_first:
            raise NotImplementedError('only the pre-LN layout of the reference configs is built')
        self.in_dim, self.d_model, self.num_heads = in_dim, d_model, num_heads
        self.ffn_dim, self.num_layers = ffn_dim, num_layers
        enc_layer = nn.TransformerEncoderLayer(d_model=d_model, nhead=num_heads,
                                               dim_feedforward=ffn_dim, norm_first=norm_first,
                                               batch_first=True)
        self.transformer_encoder = nn.TransformerEncoder(encoder_layer=enc_layer,
                                                         num_layers=num_layers,
                                                         enable_nested_tensor=False)
        self.in_proj = nn.Linear(in_dim, d_model)
        self.out_proj = nn.Linear(d_model, in_dim)
        self._packed = None

    def _apply(self, fn, *a, **k):
        self._packed = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._packed = None
        return super().load_state_dict(*a, **k)

    def _pack(self):
        if self._packed is not None:
            return self._packed
        dev = _lib.require_gpu()
        if self.in_proj.weight.device.type != 'cuda':
            raise _lib.HipLibraryError('adapter weights are on the CPU: call .cuda() first')
        keep = []

        def d(t, transpose=False):
            t = t.detach().to(dev, torch.float32)
            t = (t.t() if transpose else t).contiguous()
            keep.append(t)
            return t.data_ptr()

        layers = (_lib.EcAdapterLayer * self.num_layers)()
        for i, lyr in enumerate(self.transformer_encoder.layers):
            e = layers[i]
            e.ln1_g, e.ln1_b = d(lyr.norm1.weight), d(lyr.norm1.bias)
            e.qkv_w_t, e.qkv_b = d(lyr.self_attn.in_proj_weight, True), d(lyr.self_attn.in_proj_bias)
            e.o_w_t, e.o_b = d(lyr.self_attn.out_proj.weight, True), d(lyr.self_attn.out_proj.bias)
            e.ln2_g, e.ln2_b = d(lyr.norm2.weight), d(lyr.norm2.bias)
            e.w1_t, e.b1 = d(lyr.linear1.weight, True), d(lyr.linear1.bias)
            e.w2_t, e.b2 = d(lyr.linear2.weight, True), d(lyr.linear2.bias)
        w = _lib.EcAdapterWeights()
        w.in_dim, w.d_model, w.heads, w.ffn = self.in_dim, self.d_model, self.num_heads, self.ffn_dim
        w.layers, w.residual = self.num_layers, float(self.residual)
        w.in_w_t, w.in_b = d(self.in_proj.weight, True), d(self.in_proj.bias)
        w.out_w_t, w.out_b = d(self.out_proj.weight, True), d(self.out_proj.bias)
        w.layer = ctypes.cast(layers, ctypes.POINTER(_lib.EcAdapterLayer))
        self._packed = dict(w=w, layers=layers, keep=keep)
        return self._packed

    @torch.no_grad()
    def forward_rows(self, feats, row_idx):
        from . import torch_ops
        self._pack()
        return torch.ops.eventclip_hip.adapter_fwd(feats.float().contiguous(), row_idx.contiguous(),
                                                    torch_ops.handle_of(self))

    @torch.no_grad()
    def forward(self, feats, valid_masks):
        """feats [B, T, C], valid_masks [B, T] (True = valid view), as adapter.py:82-105.
        Padded views must hold zeros, which is what the reference's classifier passes."""
        B, T, C = feats.shape
        idx = torch.where(valid_masks, torch.arange(B * T, device=feats.device).view(B, T),
                          torch.full((B, T), -1, device=feats.device)).to(torch.int32)
        return self.forward_rows(feats.reshape(B * T, C), idx)

    @property
    def dtype(self):
        return self.in_proj.weight.dtype
