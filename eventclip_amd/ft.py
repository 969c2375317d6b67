"""Fine-tuning CLIP's vision tower on the MI355X path.

What the reference does with torch autograd around its FTCLIPClassifier -- decide which tensors of
``model.visual`` train (models/clip_cls_ft.py:44-80), inject LoRA factors into every attention block
(models/lora.py), forward / loss (:196-256), backward, Adam with one learning rate for the classifier's own
parameters and another for the tower (method.py:152-186), optionally under ``torch.cuda.amp`` (`--fp16`,
train.py:121) -- is here three layers over the C ABI:

``VisualTower``   fp32 master weights of ``clip.visual`` + the 16-bit operand copies the kernels read
                  (``ec_pack_weight16``), ``forward`` (``ec_vit_train_forward``) and ``backward``
                  (``ec_vit_train_backward``) returning gradients by state-dict name;
``LoraFactors``   the low-rank factors with the reference's key names, the merged weights they act through
                  (lora.py:138-150, :50-52); their gradients come straight from the activations inside the backward pass;
``FTTrainer``     one optimisation step: loss and feature gradients (``ec_ft_loss_grad``), the gradient scaler
                  of mixed precision (``ec_grad_unscale_check``), one all-reduce of the flat gradient buffer
                  across ranks, ``ec_adam_step`` per tensor with the warm-up + cosine schedule.
No CPU fallback: every numeric step is a HIP kernel.
"""
import ctypes
import re

import torch
import torch.distributed as dist

from . import _lib
from .train import _AGG, cosine_warmup_lr

_BLOCK = (('ln1_g', 'ln_1.weight'), ('ln1_b', 'ln_1.bias'), ('qkv_w', 'attn.in_proj_weight'),
          ('qkv_b', 'attn.in_proj_bias'), ('out_w', 'attn.out_proj.weight'), ('out_b', 'attn.out_proj.bias'),
          ('ln2_g', 'ln_2.weight'), ('ln2_b', 'ln_2.bias'), ('fc1_w', 'mlp.c_fc.weight'), ('fc1_b', 'mlp.c_fc.bias'),
          ('fc2_w', 'mlp.c_proj.weight'), ('fc2_b', 'mlp.c_proj.bias'))
_TOP = (('conv_w', 'conv1.weight'), ('cls', 'class_embedding'), ('pos', 'positional_embedding'),
        ('ln_pre_g', 'ln_pre.weight'), ('ln_pre_b', 'ln_pre.bias'), ('ln_post_g', 'ln_post.weight'),
        ('ln_post_b', 'ln_post.bias'), ('proj', 'proj'))
_MATRICES = ('qkv_w', 'out_w', 'fc1_w', 'fc2_w')


def _block_name(i, leaf):
    return f'transformer.resblocks.{i}.{leaf}'


def pack_weights16(jobs, dtype_code):
    """jobs: list of (w fp32 CUDA [rows, cols], hi, lo, hi_t) with 16-bit outputs or None, all the same shape ->
    one ``ec_pack_weight16_batched`` launch."""
    if not jobs:
        return
    rows, cols = jobs[0][0].shape
    items = (_lib.EcPackItem * len(jobs))()
    for it, (w, hi, lo, hi_t) in zip(items, jobs):
        assert w.is_cuda and w.dtype == torch.float32 and w.is_contiguous() and tuple(w.shape) == (rows, cols)
        it.w = w.data_ptr()
        it.hi = hi.data_ptr() if hi is not None else None
        it.lo = lo.data_ptr() if lo is not None else None
        it.hi_t = hi_t.data_ptr() if hi_t is not None else None
    table = device_table(items)
    rc = _lib.lib().ec_pack_weight16_batched(_lib.ptr(table), len(jobs), rows, cols, dtype_code, _lib.stream_ptr())
    _lib.check(rc, 'ec_pack_weight16_batched')
    return table


def pack_weight16(w, dtype_code, hi=None, lo=None, hi_t=None):
    """fp32 CUDA [rows, cols] -> the 16-bit copies asked for."""
    pack_weights16([(w, hi, lo, hi_t)], dtype_code)


def sgemm(a, b, out, alpha=1.0, beta=0.0):
    """out[M, N] = alpha * a[M, K] @ b[K, N] + beta * out; a / b any 2-D strided fp32 CUDA views."""
    assert a.dtype == b.dtype == out.dtype == torch.float32 and a.is_cuda and a.dim() == b.dim() == 2
    M, K = a.shape
    K2, N = b.shape
    assert K == K2 and tuple(out.shape) == (M, N) and out.stride(1) == 1
    rc = _lib.lib().ec_sgemm(_lib.ptr(a), a.stride(0), a.stride(1), _lib.ptr(b), b.stride(0), b.stride(1), M, N, K,
                             float(alpha), float(beta), _lib.ptr(out), out.stride(0), _lib.stream_ptr())
    _lib.check(rc, 'ec_sgemm')
    return out


def device_table(items):
    """ctypes array of structs -> device copy (uint8 CUDA tensor) for the batched kernels' item tables."""
    import numpy as np
    host = np.frombuffer(items, dtype=np.uint8).copy()
    return torch.from_numpy(host).cuda()


class VisualTower:
    """The vision tower of an ``eventclip_amd.clip.CLIP`` in training form.

    ``master[name]`` are the module's own fp32 parameters (state-dict names under ``visual.``), updated in
    place by the optimiser; ``effective[name]`` is what gets packed for a matrix (the master itself, or a
    LoRA-merged scratch); LayerNorm terms, biases and the embeddings are read by the kernels as fp32, straight
    from the masters."""

    def __init__(self, clip_model):
        dev = _lib.require_gpu()
        if clip_model.logit_scale.device.type != 'cuda':
            raise _lib.HipLibraryError('CLIP weights are on the CPU: call model.cuda() first (there is no CPU fallback)')
        if clip_model.image_precise:
            raise NotImplementedError('the split-precision image tower has no training form')
        self.clip, self.dev = clip_model, dev
        c = clip_model.cfg
        self.cfg = c
        self.W, self.L, self.P, self.D = c['width'], c['layers'], c['patch'], c['embed_dim']
        self.G = (c['image_size'] // c['patch']) ** 2
        self.S = self.G + 1
        self.cd = clip_model.compute_dtype
        self.code = _lib.EC_F16 if self.cd == torch.float16 else _lib.EC_BF16
        self.master = {}
        for name, p in clip_model.visual.named_parameters():
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                p.data = p.data.to(dev, torch.float32).contiguous()
            self.master[name] = p.data
        self.effective = {}
        k = 3 * self.P * self.P
        self.k, self.kpad, self.klo = k, ((2 * k + 63) // 64) * 64, ((k + 63) // 64) * 64
        W, D = self.W, self.D
        z16 = lambda *shape: torch.zeros(shape, dtype=self.cd, device=dev)      # noqa: E731
        self.packed = dict(conv=z16(W, self.kpad), conv_lo=z16(W, self.klo), conv_hi_tmp=z16(W, k),
                           conv_lo_tmp=z16(W, k), proj_lo_tmp=z16(W, D), proj_pair=z16(2, D, W))
        self.packed['proj_t'], self.packed['proj_lo_t'] = self.packed['proj_pair'][0], self.packed['proj_pair'][1]
        shapes = dict(qkv_w=(3 * W, W), out_w=(W, W), fc1_w=(4 * W, W), fc2_w=(W, 4 * W))
        for i in range(self.L):
            for f in _MATRICES:
                r, cdim = shapes[f]
                self.packed[(i, f)] = z16(r, cdim)
                self.packed[(i, f, 't')] = z16(cdim, r)
        self._pack_plans = {}
        self._build_structs()
        self.pack()
        self._ws = None
        self._tape = None
        self._grad_slots = {}
        self._grad_structs = {}

    # ---- structs ----
    def _build_structs(self):
        m, pk = self.master, self.packed
        blocks = (_lib.EcBlockWeights * self.L)()
        blocks_t = (_lib.EcBlockWeightsT * self.L)()
        for i in range(self.L):
            b = blocks[i]
            for field, leaf in _BLOCK:
                if field in _MATRICES:
                    setattr(b, field, pk[(i, field)].data_ptr())
                    setattr(blocks_t[i], field + 't', pk[(i, field, 't')].data_ptr())
                else:
                    setattr(b, field, m[_block_name(i, leaf)].data_ptr())
        v = _lib.EcVitWeights()
        c = self.cfg
        v.dtype, v.image_size, v.patch, v.width = self.code, c['image_size'], self.P, self.W
        v.layers, v.heads, v.out_dim, v.kpad = self.L, self.W // 64, self.D, self.kpad
        v.conv_w, v.conv_w_lo = pk['conv'].data_ptr(), pk['conv_lo'].data_ptr()
        v.cls, v.pos = m['class_embedding'].data_ptr(), m['positional_embedding'].data_ptr()
        v.ln_pre_g, v.ln_pre_b = m['ln_pre.weight'].data_ptr(), m['ln_pre.bias'].data_ptr()
        v.ln_post_g, v.ln_post_b = m['ln_post.weight'].data_ptr(), m['ln_post.bias'].data_ptr()
        v.proj_w, v.proj_w_lo = pk['proj_t'].data_ptr(), pk['proj_lo_t'].data_ptr()
        v.blocks = ctypes.cast(blocks, ctypes.POINTER(_lib.EcBlockWeights))
        v.precise, v.full_last_block = 0, 1
        t = _lib.EcVitTrainWeights()
        t.blocks = ctypes.cast(blocks_t, ctypes.POINTER(_lib.EcBlockWeightsT))
        t.proj = m['proj'].data_ptr()
        self._vit, self._vit_t, self._keep = v, t, (blocks, blocks_t)

    def matrix_names(self):
        out = ['conv1.weight', 'proj']
        for i in range(self.L):
            out += [_block_name(i, leaf) for f, leaf in _BLOCK if f in _MATRICES]
        return out

    def pack(self, names=None):
        """(Re)build the 16-bit operand copies of the named matrices (default: all) from ``effective``: one
        launch per distinct shape (q k v / out / c_fc / c_proj of every block together), item tables cached."""
        pk = self.packed
        todo = tuple(sorted(self.matrix_names() if names is None else set(names)))
        src = lambda n: self.effective.get(n, self.master[n])                  # noqa: E731
        plan = self._pack_plans.get(todo)
        if plan is None:
            groups = {}
            for i in range(self.L):
                for f, leaf in _BLOCK:
                    n = _block_name(i, leaf)
                    if f in _MATRICES and n in todo:
                        groups.setdefault(tuple(src(n).shape), []).append((src(n), pk[(i, f)], None, pk[(i, f, 't')]))
            if 'conv1.weight' in todo:          # -> hi / lo of the [W, 3 p^2] matrix; laid out against the patch row below
                groups.setdefault(('conv',), []).append((src('conv1.weight').reshape(self.W, self.k), pk['conv_hi_tmp'],
                                                         pk['conv_lo_tmp'], None))
            if 'proj' in todo:
                groups.setdefault(('proj',), []).append((src('proj'), None, pk['proj_lo_tmp'], pk['proj_t']))
            plan = []
            for jobs in groups.values():
                rows, cols = jobs[0][0].shape
                items = (_lib.EcPackItem * len(jobs))()
                for it, (w, hi, lo, hi_t) in zip(items, jobs):
                    assert w.is_contiguous() and w.dtype == torch.float32
                    it.w = w.data_ptr()
                    it.hi = hi.data_ptr() if hi is not None else None
                    it.lo = lo.data_ptr() if lo is not None else None
                    it.hi_t = hi_t.data_ptr() if hi_t is not None else None
                plan.append((device_table(items), len(jobs), rows, cols))
            self._pack_plans[todo] = plan
        for table, n, rows, cols in plan:
            rc = _lib.lib().ec_pack_weight16_batched(_lib.ptr(table), n, rows, cols, self.code, _lib.stream_ptr())
            _lib.check(rc, 'ec_pack_weight16_batched')
        if 'conv1.weight' in todo:
            pk['conv'][:, :self.k] = pk['conv_hi_tmp']                          # [w_hi | w_hi | 0]
            pk['conv'][:, self.k:2 * self.k] = pk['conv_hi_tmp']
            pk['conv_lo'][:, :self.k] = pk['conv_lo_tmp']                       # [w_lo | 0]
        if 'proj' in todo:
            pk['proj_lo_t'].copy_(pk['proj_lo_tmp'].t())
        self.clip._packed = None       # the inference copies of clip.py are stale once a master moved

    # ---- passes ----
    def _workspace(self, n):
        need = int(_lib.lib().ec_vit_train_workspace_bytes(ctypes.byref(self._vit), n))
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty((need,), dtype=torch.uint8, device=self.dev)
        return self._ws

    def forward(self, patches):
        """patches: 16-bit CUDA [N, G, kpad] -> fp32 features [N, D]; keeps the tape for ``backward``."""
        assert patches.is_cuda and patches.dtype == self.cd and patches.is_contiguous()
        n = patches.shape[0]
        assert tuple(patches.shape) == (n, self.G, self.kpad), f'patches {tuple(patches.shape)}'
        ws = self._workspace(n)
        feats = torch.empty((n, self.D), dtype=torch.float32, device=self.dev)
        rc = _lib.lib().ec_vit_train_forward(ctypes.byref(self._vit), _lib.ptr(patches), n, _lib.ptr(feats),
                                             _lib.ptr(ws), ws.numel(), _lib.stream_ptr())
        _lib.check(rc, 'ec_vit_train_forward')
        self._tape = (patches, n)
        return feats

    def encode_patches(self, patches):
        """Inference through the SAME (possibly LoRA-merged) operand copies, every token of the last block:
        evaluation in the middle of a fine-tuning run."""
        n = patches.shape[0]
        need = int(_lib.lib().ec_vit_workspace_bytes(ctypes.byref(self._vit), min(n, 256)))
        ws = torch.empty((need,), dtype=torch.uint8, device=self.dev)
        feats = torch.empty((n, self.D), dtype=torch.float32, device=self.dev)
        rc = _lib.lib().ec_vit_encode(ctypes.byref(self._vit), _lib.ptr(patches), n, _lib.ptr(feats), _lib.ptr(ws),
                                      ws.numel(), min(n, 256), _lib.stream_ptr())
        _lib.check(rc, 'ec_vit_encode')
        return feats

    def grad_buffer(self, want):
        """One flat fp32 buffer with a slot per wanted gradient (a single all-reduce / unscale covers it)."""
        key = tuple(want)
        if key not in self._grad_slots:
            sizes = [self.master[n].numel() for n in want]
            flat = torch.zeros((max(sum(sizes), 4),), dtype=torch.float32, device=self.dev)
            views, off = {}, 0
            for n, sz in zip(want, sizes):
                views[n] = flat[off:off + sz].view(self.master[n].shape)
                off += sz
            self._grad_slots[key] = (flat, views)
        return self._grad_slots[key]

    def _grad_struct(self, want):
        """(EcVitGrads over the flat buffer's views, views, flat, {name: (offset, numel)}) for a want list, cached."""
        key = tuple(want)
        if key not in self._grad_structs:
            flat, views = self.grad_buffer(want)
            bg = (_lib.EcBlockGrads * self.L)()
            g = _lib.EcVitGrads()
            top = {leaf: f for f, leaf in _TOP}
            blk = {leaf: f for f, leaf in _BLOCK}
            spans, off = {}, 0
            for name, t in views.items():
                spans[name] = (off, t.numel())
                off += t.numel()
                if name in top:
                    setattr(g, top[name], t.data_ptr())
                else:
                    m = re.match(r'^transformer\.resblocks\.(\d+)\.(.+)$', name)
                    if not m or m.group(2) not in blk:
                        raise KeyError(f'{name!r} is not a parameter of the vision tower')
                    setattr(bg[int(m.group(1))], blk[m.group(2)], t.data_ptr())
            g.blocks = ctypes.cast(bg, ctypes.POINTER(_lib.EcBlockGrads))
            self._grad_structs[key] = (g, bg, views, flat, spans)
        g, _, views, flat, spans = self._grad_structs[key]
        return g, views, flat, spans

    def backward(self, d_feats, want, lora=None, stages=None):
        """d_feats fp32 [N, D] -> {name: gradient} for the state-dict names in ``want`` (views of one flat
        buffer, also returned).  lora: an ``EcVitLora`` whose factor gradients are written alongside.
        stages = (begin, end): only that part of the pass (0 = head, 1 .. L = blocks L - 1 .. 0, L + 1 = the
        embedding; ``ec_vit_train_backward_stages``), for a gradient exchange overlapped with the rest."""
        assert self._tape is not None, 'backward without a forward'
        patches, n = self._tape
        assert d_feats.is_cuda and d_feats.dtype == torch.float32 and tuple(d_feats.shape) == (n, self.D)
        assert d_feats.is_contiguous()
        g, views, flat, _ = self._grad_struct(want)
        sb, se = (0, self.L + 2) if stages is None else stages
        ws = self._workspace(n)
        rc = _lib.lib().ec_vit_train_backward_stages(ctypes.byref(self._vit), ctypes.byref(self._vit_t), _lib.ptr(patches),
                                                     n, _lib.ptr(d_feats), ctypes.byref(g),
                                                     ctypes.byref(lora) if lora is not None else None, int(sb), int(se),
                                                     _lib.ptr(ws), ws.numel(), _lib.stream_ptr())
        _lib.check(rc, 'ec_vit_train_backward_stages')
        return views, flat

    def canonical(self, names):
        """The names in gradient-buffer order: embedding | blocks 0 .. L-1 | head, so that the part of the buffer a
        stretch of the backward pass completes is one contiguous slice."""
        emb = ('conv1.weight', 'class_embedding', 'positional_embedding', 'ln_pre.weight', 'ln_pre.bias')
        head = ('ln_post.weight', 'ln_post.bias', 'proj')
        leaf = {lf: i for i, (_, lf) in enumerate(_BLOCK)}

        def rank(n):
            if n in emb:
                return (0, 0, emb.index(n))
            if n in head:
                return (2, 0, head.index(n))
            m = re.match(r'^transformer\.resblocks\.(\d+)\.(.+)$', n)
            if not m or m.group(2) not in leaf:
                raise KeyError(f'{n!r} is not a parameter of the vision tower')
            return (1, int(m.group(1)), leaf[m.group(2)])
        return sorted(names, key=rank)

    def bucket_plan(self, want, blocks_per_bucket=4):
        """[(stage_begin, stage_end, flat_lo, flat_hi)]: the backward pass cut into the head, groups of
        ``blocks_per_bucket`` blocks (last block first) and the embedding, each with the contiguous slice of the flat
        gradient buffer it completes (the buffer follows the state dict's order: embedding | blocks 0 .. L-1 | head)."""
        assert list(want) == self.canonical(want), 'bucket_plan needs the names in canonical() order'
        _, _, _, spans = self._grad_struct(want)

        def span(pred):
            hit = [spans[n] for n in want if pred(n)]
            return (min(o for o, _ in hit), max(o + c for o, c in hit)) if hit else (0, 0)

        def block_of(n):
            m = re.match(r'^transformer\.resblocks\.(\d+)\.', n)
            return int(m.group(1)) if m else None
        L, plan = self.L, []
        plan.append((0, 1) + span(lambda n: n in ('ln_post.weight', 'ln_post.bias', 'proj')))
        hi = L - 1
        while hi >= 0:
            lo = max(hi - blocks_per_bucket + 1, 0)
            plan.append((L - hi, L - lo + 1) + span(lambda n, lo=lo, hi=hi: block_of(n) is not None and lo <= block_of(n) <= hi))
            hi = lo - 1
        plan.append((L + 1, L + 2) + span(lambda n: block_of(n) is None and n not in ('ln_post.weight', 'ln_post.bias', 'proj')))
        return plan


# ---- which tensors train, LoRA ------------------------------------------------------------------------
def parse_lora(spec):
    """models/lora.py:352-365 -> None or (r, lora_k, lora_o)."""
    if isinstance(spec, str):
        if not ('q' in spec and 'v' in spec):
            raise AssertionError("LoRA spec must name q and v ('qv-R', 'qkv-R', 'qkvo-R')")
        return int(spec.split('-')[-1]), 'k' in spec, 'o' in spec
    if spec is None or spec <= 0:
        return None
    return int(spec), True, False


def trainable_visual(names, clip_dict):
    """The parameters of ``model.visual`` the reference leaves trainable (clip_cls_ft.py:44-80), LoRA
    factors not included (they are not module parameters here)."""
    g = clip_dict.get
    picked = set()
    if g('only_conv1', False):
        picked.add('conv1.weight')
    if g('only_bias', False):
        picked |= {n for n in names if 'bias' in n}
    if g('only_ln', False):
        picked |= {n for n in names if re.search(r'(^|\.)ln_(pre|post|1|2)\.', n)}
    if g('only_cls_fc', False):
        picked.add('proj')
    if g('only_cls_token', False):
        picked.add('class_embedding')
    if parse_lora(g('lora', -1)) is None and not picked:
        picked = set(names)
    return [n for n in names if n in picked]


class LoraFactors:
    """Low-rank factors of every attention block, keyed like the reference's injected modules
    (``...attn.in_proj_weight.lora_down_q`` / ``lora_up_q`` / ``_k`` / ``_v``, ``...attn.out_proj.lora_down.weight``
    / ``lora_up.weight``), initialised as lora.py:8-11 (down ~ N(0, 1 / r), up = 0)."""

    def __init__(self, tower, spec):
        self.tower = tower
        self.r, self.lora_k, self.lora_o = parse_lora(spec)
        W, dev = tower.W, tower.dev
        self.params = {}
        for i in range(tower.L):
            pre = _block_name(i, 'attn')
            for nm in ('q', 'v') + (('k',) if self.lora_k else ()):
                self.params[f'{pre}.in_proj_weight.lora_down_{nm}'] = (torch.randn(self.r, W) / self.r).to(dev)
                self.params[f'{pre}.in_proj_weight.lora_up_{nm}'] = torch.zeros(W, self.r, device=dev)
            if self.lora_o:
                self.params[f'{pre}.out_proj.lora_down.weight'] = (torch.randn(self.r, W) / self.r).to(dev)
                self.params[f'{pre}.out_proj.lora_up.weight'] = torch.zeros(W, self.r, device=dev)
        self.merged_names = []
        for i in range(tower.L):
            names = [_block_name(i, 'attn.in_proj_weight')] + ([_block_name(i, 'attn.out_proj.weight')] if self.lora_o else [])
            for n in names:
                tower.effective[n] = torch.empty_like(tower.master[n])
                self.merged_names.append(n)

    def projections(self):
        """(block, row block of in_proj or None for out_proj, down key, up key) of every injected projection."""
        out = []
        for i in range(self.tower.L):
            pre = _block_name(i, 'attn')
            for j, nm in enumerate('qkv'):
                if f'{pre}.in_proj_weight.lora_up_{nm}' in self.params:
                    out.append((i, j, f'{pre}.in_proj_weight.lora_down_{nm}', f'{pre}.in_proj_weight.lora_up_{nm}'))
            if self.lora_o:
                out.append((i, None, pre + '.out_proj.lora_down.weight', pre + '.out_proj.lora_up.weight'))
        return out

    def bind(self, factor_grads):
        """Build the item tables and the gradient struct once: every pointer (masters, merged scratch, the
        factors, their 16-bit operand copies and gradients) is stable for the life of the trainer."""
        t, W, r = self.tower, self.tower.W, self.r
        proj = self.projections()
        items = (_lib.EcLoraItem * len(proj))()
        rp = (r + 15) // 16 * 16
        # 16-bit operand copies for the backward pass: down [r, W] and up^T [r, W], rows padded to 16 with zeros
        self.down16 = {kd: torch.zeros((rp, W), dtype=t.cd, device=t.dev) for _, _, kd, _ in proj}
        self.up16t = {ku: torch.zeros((rp, W), dtype=t.cd, device=t.dev) for _, _, _, ku in proj}
        blocks = (_lib.EcBlockLora * t.L)()
        for it, (i, j, kd, ku) in zip(items, proj):
            pre = _block_name(i, 'attn')
            name = pre + ('.in_proj_weight' if j is not None else '.out_proj.weight')
            rows = slice(j * W, (j + 1) * W) if j is not None else slice(None)
            it.base = t.master[name][rows].data_ptr()
            it.out = t.effective[name][rows].data_ptr()
            it.up, it.down = self.params[ku].data_ptr(), self.params[kd].data_ptr()
            slot = 3 if j is None else j
            blocks[i].down16[slot], blocks[i].up16_t[slot] = self.down16[kd].data_ptr(), self.up16t[ku].data_ptr()
            blocks[i].d_up[slot], blocks[i].d_down[slot] = factor_grads[ku].data_ptr(), factor_grads[kd].data_ptr()
        self._n_items = len(proj)
        self._items = device_table(items)
        self._blocks = blocks
        self.struct = _lib.EcVitLora()
        self.struct.rank = r
        self.struct.blocks = ctypes.cast(blocks, ctypes.POINTER(_lib.EcBlockLora))
        self._pack_down = [(self.params[kd], self.down16[kd], None, None) for _, _, kd, _ in proj]
        self._pack_up = [(self.params[ku], None, None, self.up16t[ku]) for _, _, _, ku in proj]
        self._pack_tables = None
        if not self.lora_k:       # the k rows of in_proj carry no factors: their merged rows are the base rows
            for i in range(t.L):
                n = _block_name(i, 'attn.in_proj_weight')
                t.effective[n][W:2 * W].copy_(t.master[n][W:2 * W])

    def merge(self):
        """effective = base + up @ down for every projection (lora.py:138-150, :50-52) and the repack of those
        matrices; the factors' own 16-bit copies (the backward pass's operands) follow."""
        t = self.tower
        rc = _lib.lib().ec_lora_merge_batched(_lib.ptr(self._items), self._n_items, t.W, t.W, self.r, _lib.stream_ptr())
        _lib.check(rc, 'ec_lora_merge_batched')
        t.pack(self.merged_names)
        if self._pack_tables is None:
            def table(jobs):
                items = (_lib.EcPackItem * len(jobs))()
                for it, (w, hi, lo, hi_t) in zip(items, jobs):
                    it.w = w.data_ptr()
                    it.hi = hi.data_ptr() if hi is not None else None
                    it.hi_t = hi_t.data_ptr() if hi_t is not None else None
                return device_table(items)
            self._pack_tables = (table(self._pack_down), table(self._pack_up))
        for tbl, (rows, cols) in zip(self._pack_tables, ((self.r, t.W), (t.W, self.r))):
            rc = _lib.lib().ec_pack_weight16_batched(_lib.ptr(tbl), self._n_items, rows, cols, t.code, _lib.stream_ptr())
            _lib.check(rc, 'ec_pack_weight16_batched')

    def state_dict_entries(self):
        """The tower's attention entries as the reference's LoRA-injected modules name them."""
        t, out = self.tower, {}
        for i in range(t.L):
            pre = _block_name(i, 'attn')
            out[pre + '.in_proj_weight.merged_proj'] = t.master[pre + '.in_proj_weight']
            if self.lora_o:
                out[pre + '.out_proj.linear.weight'] = t.master[pre + '.out_proj.weight']
                out[pre + '.out_proj.linear.bias'] = t.master[pre + '.out_proj.bias']
        out.update(self.params)
        return out


class GradScaler:
    """torch.cuda.amp.GradScaler's policy (what `--fp16` gives the reference, train.py:121): gradients flow
    scaled, a step with a non-finite gradient is skipped and halves the scale, ``growth_interval`` clean steps
    double it."""

    def __init__(self, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, enabled=True):
        self.scale = float(init_scale) if enabled else 1.0
        self.growth_factor, self.backoff_factor = growth_factor, backoff_factor
        self.growth_interval, self.enabled, self._good = growth_interval, enabled, 0

    def update(self, found_inf):
        if not self.enabled:
            return
        if found_inf:
            self.scale *= self.backoff_factor
            self._good = 0
        else:
            self._good += 1
            if self._good == self.growth_interval:
                self.scale *= self.growth_factor
                self._good = 0


def ft_loss_grad(img_feats, valid, labels, text_param, logit_scale, agg='mean', use_probs_loss=False, grad_scale=1.0,
                 want_text_grad=True, text_grad_out=None, row_idx=None, step_scalars=None):
    """The classifier head in train mode (``ec_ft_loss_grad``).  img_feats fp32 CUDA: [B, T, D] with zero rows on
    invalid views, or -- with ``row_idx`` int32 [B, T] (-1 = invalid) -- compact [Nv, D] over the valid views.
    Returns (loss, d loss / d img_feats * grad_scale in the same layout, d loss / d text_param or None,
    aggregated logits [B, K])."""
    dev = _lib.require_gpu()
    if agg not in _AGG:
        raise NotImplementedError(f'agg_func {agg!r}: the reference trains with sum / mean')
    f = img_feats.float().contiguous()
    B, T = valid.shape
    D = f.shape[-1]
    if row_idx is None:
        assert tuple(f.shape) == (B, T, D)
    else:
        row_idx = row_idx.to(torch.int32).contiguous()
        assert f.dim() == 2 and tuple(row_idx.shape) == (B, T)
    t = text_param.detach().float().contiguous()
    K = t.shape[0]
    v8 = valid.to(torch.uint8).contiguous()
    lab = labels.to(torch.int32).contiguous()
    need = int(_lib.lib().ec_fs_text_train_workspace_bytes(B, T, D, K)) + max(B * T, K) * D * 4
    ws = torch.empty((need,), dtype=torch.uint8, device=dev)
    loss = torch.empty((1,), dtype=torch.float32, device=dev)
    gimg = torch.zeros_like(f)
    gtext = None
    if want_text_grad:
        gtext = text_grad_out if text_grad_out is not None else torch.empty((K, D), dtype=torch.float32, device=dev)
        assert gtext.is_contiguous() and tuple(gtext.shape) == (K, D) and gtext.dtype == torch.float32
    logits = torch.empty((B, K), dtype=torch.float32, device=dev)
    rc = _lib.lib().ec_ft_loss_grad(_lib.ptr(f), _lib.ptr(row_idx), _lib.ptr(v8), _lib.ptr(lab), _lib.ptr(t), B, T, D, K,
                                    float(logit_scale), _AGG[agg], int(bool(use_probs_loss)), float(grad_scale),
                                    _lib.ptr(step_scalars), _lib.ptr(loss), _lib.ptr(gtext), _lib.ptr(gimg),
                                    _lib.ptr(logits), _lib.ptr(ws),
                                    ws.numel(), _lib.stream_ptr())
    _lib.check(rc, 'ec_ft_loss_grad')
    return loss[0], gimg, gtext, logits


class FTTrainer:
    """One optimisation step of the reference's fine-tuning (train.py / method.py with ``model = 'FTCLIP'``)
    for an ``eventclip_amd.clip_cls_ft.FTCLIPClassifier``.

    ``lr`` drives the classifier's own parameters (``text_feats``), ``clip_lr`` those under ``model.visual``
    (method.py:166-178); both follow the warm-up + cosine schedule (min = max / 100, :179-186).
    ``mixed_precision`` keeps torch.cuda.amp.GradScaler's policy around the 16-bit gradient path."""

    def __init__(self, classifier, lr, clip_lr=None, total_steps=1000, warmup_steps_pct=0.05, optimizer='Adam',
                 betas=(0.9, 0.999), eps=1e-8, weight_decay=0., mixed_precision=True, init_scale=65536.0,
                 growth_interval=2000, blocks_per_bucket=4, graph=False, graph_warmup=2):
        if optimizer.lower() not in ('adam', 'adamw'):
            raise ValueError('Should use Adam or AdamW optimizer!')                     # method.py:160
        assert weight_decay == 0.                                                       # method.py:161
        self.clf = classifier
        self.tower = VisualTower(classifier.model)
        cd = classifier.clip_dict
        names = list(self.tower.master)
        self.visual_train = trainable_visual(names, cd)
        self.lora = LoraFactors(self.tower, cd.get('lora', -1)) if parse_lora(cd.get('lora', -1)) else None
        self.tensors = {}
        if classifier.prompt_tuning:
            self.tensors['text_feats'] = classifier.text_feats.data
        for n in self.visual_train:
            self.tensors['model.visual.' + n] = self.tower.master[n]
        if self.lora:
            for n, p in self.lora.params.items():
                self.tensors['model.visual.' + n] = p
        self._broadcast_initial_state()
        self.lr = float(lr)
        self.clip_lr = float(lr if clip_lr is None else clip_lr)
        self.total_steps = int(total_steps)
        self.warmup_steps = warmup_steps_pct * self.total_steps
        self.betas, self.eps = betas, float(eps)
        self.state = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in self.tensors.items()}
        self.scaler = GradScaler(init_scale, growth_interval=growth_interval, enabled=mixed_precision)
        self.steps = 0            # scheduler steps (every call, as the reference steps its scheduler)
        self.opt_steps = 0        # optimiser steps actually taken (a skipped step does not advance Adam)
        # what the tower has to differentiate (the LoRA factors' gradients come out of the same pass, separately)
        self.want = self.tower.canonical(self.visual_train)
        self._found = torch.zeros((1,), dtype=torch.int32, device=self.tower.dev)
        self._lora_grads = {}
        if self.lora:                                   # the factor gradients share one flat buffer too
            total = sum(p.numel() for p in self.lora.params.values())
            self._lora_flat = torch.zeros((total,), dtype=torch.float32, device=self.tower.dev)
            off = 0
            for n, p in self.lora.params.items():
                self._lora_grads[n] = self._lora_flat[off:off + p.numel()].view(p.shape)
                off += p.numel()
        # gradients live at fixed addresses: views of the tower's flat buffer, the factor gradients, text_feats'
        self._grads = {}
        self._flat, views = self.tower.grad_buffer(self.want)
        for n in self.visual_train:
            self._grads['model.visual.' + n] = views[n]
        if self.lora:
            self.lora.bind(self._lora_grads)
            for n, g in self._lora_grads.items():
                self._grads['model.visual.' + n] = g
            self.lora.merge()
        if classifier.prompt_tuning:
            self._grads['text_feats'] = torch.zeros_like(classifier.text_feats.data)
        items = (_lib.EcAdamItem * len(self.tensors))()
        for it, (k, p) in zip(items, self.tensors.items()):
            assert p.is_contiguous() and self._grads[k].is_contiguous()
            m, v = self.state[k]
            it.param, it.grad, it.exp_avg, it.exp_avg_sq = p.data_ptr(), self._grads[k].data_ptr(), m.data_ptr(), v.data_ptr()
            it.n, it.group = p.numel(), int(k.startswith('model.visual.'))
        self._adam_items = device_table(items) if len(self.tensors) else None
        self._adam_max = max([p.numel() for p in self.tensors.values()] + [0])
        # the scaler's verdict on a step is read back asynchronously (pinned host word + event) and applied
        # before the next step needs the scale: the host never waits for the step it has just queued
        matrices = set(self.tower.matrix_names())
        self._moved = [n for n in self.visual_train if n in matrices]
        self._buckets = self.tower.bucket_plan(self.want, blocks_per_bucket) if self.want else []
        self._found_host = torch.zeros((1,), dtype=torch.int32).pin_memory()
        self._scalars_dev = torch.zeros((_lib.EC_STEP_COUNT,), dtype=torch.float32, device=self.tower.dev)
        self.graph, self.graph_warmup, self._graph, self._static = bool(graph), int(graph_warmup), None, None
        self._fixed_text = None
        self._pending = None
        classifier._tower, classifier._trainer = self.tower, self
        self.last = {}

    def _broadcast_initial_state(self):
        """Under torch.distributed every rank starts from rank 0's tensors, as torch's DistributedDataParallel
        does at construction (the reference wraps the model in it, nerv / train.py --ddp): the LoRA down factors
        are drawn from each rank's own generator (lora.py:8-11), and ranks usually seed differently for their
        data augmentation -- averaging gradients over replicas that started apart lets them drift apart for good.
        Masters that do not train are broadcast too (a rank may have built them differently); the 16-bit operand
        copies are rebuilt from what arrived."""
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return
        seen = set()
        for t in list(self.tensors.values()) + list(self.tower.master.values()):
            if t.data_ptr() in seen:
                continue
            seen.add(t.data_ptr())
            dist.broadcast(t, src=0)
        self.tower.pack()

    def trainable_names(self):
        return sorted(self.tensors)

    # ---- optimiser state: what a resumed run needs beyond the parameters (torch.optim.Adam.state_dict + the
    # scheduler's step count + GradScaler.state_dict in the reference's checkpoints) ----
    def state_dict(self):
        self.resolve()
        return {'exp_avg': {k: m.clone() for k, (m, _) in self.state.items()},
                'exp_avg_sq': {k: v.clone() for k, (_, v) in self.state.items()},
                'steps': self.steps, 'opt_steps': self.opt_steps,
                'scaler': {'scale': self.scaler.scale, 'good_steps': self.scaler._good}}

    def load_state_dict(self, sd):
        self.resolve()
        missing = set(self.state) ^ set(sd['exp_avg'])
        if missing:
            raise KeyError(f'optimiser state does not match the trainable tensors: {sorted(missing)[:3]} ...')
        for k, (m, v) in self.state.items():          # in place: the Adam item table holds these addresses
            m.copy_(sd['exp_avg'][k])
            v.copy_(sd['exp_avg_sq'][k])
        self.steps, self.opt_steps = int(sd['steps']), int(sd['opt_steps'])
        self.scaler.scale, self.scaler._good = float(sd['scaler']['scale']), int(sd['scaler']['good_steps'])
        self._graph = None                             # a captured step baked the old scalars' addresses' contents in

    def load_visual_state_dict(self, vis, strict=True):
        """``model.visual.*`` of a checkpoint (keys without that prefix) INTO the attached trainer: the masters in
        place, the LoRA factors into ``lora.params`` (``merged_proj`` / ``linear.*`` are the FROZEN base weights
        there, lora.py:138-150, and go to the masters unmerged), then the merge and the repack of the 16-bit
        operand copies.  FTCLIPClassifier.load_state_dict routes here while a trainer is attached."""
        self.resolve()
        t = self.tower
        vis = dict(vis)
        src_of = {name: name for name in t.master}          # checkpoint key of every master
        if self.lora:
            for i in range(t.L):
                pre = _block_name(i, 'attn')
                src_of[pre + '.in_proj_weight'] = pre + '.in_proj_weight.merged_proj'
                if self.lora.lora_o:
                    src_of[pre + '.out_proj.weight'] = pre + '.out_proj.linear.weight'
                    src_of[pre + '.out_proj.bias'] = pre + '.out_proj.linear.bias'
        targets = [(m, src_of[name]) for name, m in t.master.items()]
        if self.lora:
            targets += [(p, k) for k, p in self.lora.params.items()]
        for dst, key in targets:
            if key in vis:
                dst.copy_(torch.as_tensor(vis[key]).to(dst.device, dst.dtype))      # in place: addresses are baked in
            elif strict:
                raise KeyError(f'checkpoint lacks model.visual.{key}')
        if self.lora and not self.lora.lora_k:    # the k rows carry no factors: their merged rows are the base rows
            W = t.W
            for i in range(t.L):
                n = _block_name(i, 'attn.in_proj_weight')
                t.effective[n][W:2 * W].copy_(t.master[n][W:2 * W])
        t.pack()
        if self.lora:
            self.lora.merge()
        t.clip._packed = None                          # the inference copies are rebuilt from the masters on demand
        self._graph = None

    def _patches(self, data_dict):
        """-> patches [Nv, G, kpad] of the valid views, valid [B, T], row_idx [B, T] (no host synchronisation when
        the batch comes with its own 'patches' + 'row_idx', e.g. from Event2ImagePipeline)."""
        valid = data_dict['valid_mask'].to(self.tower.dev)
        if 'patches' in data_dict:
            return data_dict['patches'], valid, data_dict['row_idx']
        flat = valid.reshape(-1)
        row_idx = torch.where(flat, torch.cumsum(flat.int(), 0) - 1, torch.full_like(flat, -1, dtype=torch.int64))
        row_idx = row_idx.to(torch.int32).reshape(valid.shape)
        imgs = data_dict['img']
        t = self.tower
        x = imgs[valid].to(t.dev, torch.float32).contiguous()
        R = t.cfg['image_size']
        patches = torch.empty((x.shape[0], t.G, t.kpad), dtype=t.cd, device=t.dev)
        rc = _lib.lib().ec_patchify(_lib.ptr(x), x.shape[0], R, t.P, t.kpad, _lib.ptr(patches), t.code,
                                    _lib.stream_ptr())
        _lib.check(rc, 'ec_patchify')
        return patches, valid, row_idx

    def resolve(self):
        """Apply the gradient scaler's verdict on the last queued step (waits for that step if it is still
        running).  Called by the next ``step``; call it yourself before reading ``last['skipped']``,
        ``opt_steps`` or ``scaler.scale`` right after a step."""
        if self._pending is None:
            return
        self._pending.synchronize()
        self._pending = None
        found = bool(int(self._found_host[0]))
        if not found:
            self.opt_steps += 1
        self.scaler.update(found)
        self.last['skipped'] = found

    def _scalars(self, S, world):
        """The per-step scalars as the kernels read them from device memory (EC_STEP_*)."""
        lr = cosine_warmup_lr(self.steps, self.total_steps, self.lr, self.lr / 100., self.warmup_steps)
        clip_lr = cosine_warmup_lr(self.steps, self.total_steps, self.clip_lr, self.clip_lr / 100., self.warmup_steps)
        t = self.opt_steps + 1
        h = [0.0] * _lib.EC_STEP_COUNT
        h[_lib.EC_STEP_LR0], h[_lib.EC_STEP_LR1] = lr, clip_lr
        h[_lib.EC_STEP_BC1] = 1.0 - self.betas[0] ** t
        h[_lib.EC_STEP_BC2_SQRT] = (1.0 - self.betas[1] ** t) ** 0.5
        h[_lib.EC_STEP_GRAD_SCALE], h[_lib.EC_STEP_INV_SCALE] = S, 1.0 / (S * world)
        # a fresh pinned block per step: the copy is asynchronous and the host may be a step ahead of the GPU
        self._scalars_dev.copy_(torch.tensor(h, dtype=torch.float32).pin_memory(), non_blocking=True)

    def _body(self, patches, valid, labels, row_idx, ddp, world):
        """Everything of a step that runs on the GPU, in stream order, with no host decision in it: the forward pass,
        the head, the backward pass (with its gradient exchange), unscale + check, Adam, re-merge / re-pack.  The
        per-step scalars come from ``self._scalars_dev``, so the same sequence can be replayed from a hipGraph."""
        clf, t = self.clf, self.tower
        sc = self._scalars_dev
        feats = t.forward(patches)                                 # [Nv, D]; scattered by row_idx inside the head
        text = clf.text_feats.data if clf.prompt_tuning else self._fixed_text
        loss, gimg, gtext, logits = ft_loss_grad(feats, valid, labels, text, clf.logit_scale, clf.agg_func,
                                                 clf.use_probs_loss, want_text_grad=clf.prompt_tuning,
                                                 text_grad_out=self._grads.get('text_feats'), row_idx=row_idx,
                                                 step_scalars=sc)
        through_tower = bool(self.want) or self.lora is not None
        check = through_tower and self.scaler.enabled
        if through_tower:
            lora_struct = self.lora.struct if self.lora else None
            if ddp and self.want:
                # DistributedDataParallel's bucketed exchange: a few blocks of the backward pass, then the all-reduce
                # of the slice of the flat buffer they completed starts on the collective's stream (RCCL over xGMI)
                # while the next blocks compute; everything is waited for before the optimiser reads it
                works, flat = [], self._flat
                for sb, se, lo, hi in self._buckets:
                    t.backward(gimg, self.want, lora_struct, stages=(sb, se))
                    if hi > lo:
                        works.append(dist.all_reduce(flat[lo:hi], async_op=True))
                for wk in works:
                    wk.wait()
            else:
                _, flat = t.backward(gimg, self.want, lora_struct)
            self._found.zero_()
            for buf in ([flat] if self.want else []) + ([self._lora_flat] if self.lora else []):
                if ddp and buf is not flat:
                    dist.all_reduce(buf)              # the LoRA factors' gradients: one small collective
                rc = _lib.lib().ec_grad_unscale_check(_lib.ptr(buf), buf.numel(), 1.0, _lib.ptr(self._found), _lib.ptr(sc),
                                                      _lib.stream_ptr())
                _lib.check(rc, 'ec_grad_unscale_check')
        if clf.prompt_tuning and ddp:
            dist.all_reduce(gtext)
            gtext /= world                      # (not scaled: the head's text gradient never sees the loss scale)
        if self._adam_items is not None:
            rc = _lib.lib().ec_adam_step_multi(_lib.ptr(self._adam_items), len(self.tensors), self._adam_max, 0., 0.,
                                               self.betas[0], self.betas[1], self.eps, 0., 0,
                                               _lib.ptr(self._found) if check else None, _lib.ptr(sc), _lib.stream_ptr())
            _lib.check(rc, 'ec_adam_step_multi')
            if self.lora:
                self.lora.merge()
            if self._moved:
                t.pack(self._moved)
            else:
                t.clip._packed = None
        if check:
            self._found_host.copy_(self._found, non_blocking=True)
        return loss, logits, feats

    @torch.no_grad()
    def step(self, data_dict):
        """data_dict: 'img' [B, T, 3, R, R] (or 'patches' [Nv, G, kpad] of the valid views in (b, t) order with
        'row_idx'), 'valid_mask' [B, T], 'label' [B].  Returns the loss (0-dim CUDA tensor).

        With ``graph=True`` the GPU work of a step is recorded once into a hipGraph (after ``graph_warmup`` eager
        steps, for batches that arrive as 'patches' + 'row_idx' with every step the same shapes and no process
        group) and replayed: the ~700 launches of a step cost the host one call."""
        clf, t = self.clf, self.tower
        patches, valid, row_idx = self._patches(data_dict)
        labels = data_dict['label'].to(t.dev)
        ddp = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        world = dist.get_world_size() if ddp else 1
        if not clf.prompt_tuning:
            self._fixed_text = clf.get_text_feats().float()
        self.resolve()                                # the previous step's verdict: this step's scale
        self._scalars(self.scaler.scale, world)
        check = (bool(self.want) or self.lora is not None) and self.scaler.enabled
        use_graph = self.graph and not ddp and 'patches' in data_dict
        if use_graph and self._graph is None and self.steps >= self.graph_warmup:
            self._capture(patches, valid, labels, row_idx)
        if use_graph and self._graph is not None and self._static['patches'].shape == patches.shape and \
                self._static['valid'].shape == valid.shape:
            st = self._static
            st['patches'].copy_(patches), st['valid'].copy_(valid), st['labels'].copy_(labels), st['row_idx'].copy_(row_idx)
            self._graph.replay()
            loss, logits, feats = st['out']
        else:
            loss, logits, feats = self._body(patches, valid, labels, row_idx, ddp, world)
        self.last = dict(logits=logits, feats=feats, grads=self._grads, skipped=False)
        self.steps += 1
        if hasattr(clf, '_invalidate_text_cache') and self._adam_items is not None:
            clf._invalidate_text_cache()
        if check:
            self._pending = torch.cuda.Event()
            self._pending.record()
        else:
            self.opt_steps += 1
        return loss

    def _capture(self, patches, valid, labels, row_idx):
        """Record ``_body`` over static input buffers into a hipGraph (torch.cuda.CUDAGraph = hipGraph on ROCm)."""
        st = dict(patches=patches.clone(), valid=valid.clone(), labels=labels.clone(), row_idx=row_idx.clone())
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            st['out'] = self._body(st['patches'], st['valid'], st['labels'], st['row_idx'], False, 1)
        # (the capture does not execute: the recorded step runs at the first replay)
        self._graph, self._static = g, st

    def visual_state_dict(self):
        """``model.visual.*`` as the reference's checkpoint holds it (LoRA keys when LoRA is on)."""
        t = self.tower
        out = dict(t.master)
        if self.lora:
            for i in range(t.L):
                pre = _block_name(i, 'attn')
                out.pop(pre + '.in_proj_weight')
                if self.lora.lora_o:
                    out.pop(pre + '.out_proj.weight')
                    out.pop(pre + '.out_proj.bias')
            out.update(self.lora.state_dict_entries())
        return {'model.visual.' + k: v for k, v in out.items()}
