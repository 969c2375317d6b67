"""Time one fine-tuning step of the vision tower on the MI355X (the reference's FTCLIP configs:
configs/ftclip/ft_text_fsclip_nin_params*.py -- ViT-L/14, N-ImageNet geometry, 128 // 4 gpus = 32 samples x
2 views per GPU, K = 1000, Adam, `lora='qkvo-16'` or every visual parameter).

    python tools/bench_ft.py [--arch ViT-L/14] [--samples 32] [--views 2] [--classes 1000] [--steps 5]
                             [--modes lora,full,bias] [--dtype float16]

Prints one JSON line per mode: ms per step, frames/s, the split forward / loss / backward / update, the
algorithmic flops (3 x the forward's 2 M N K for a full step; LoRA skips the MLP's weight gradients) and the
rate they amount to, and `cpu_baseline`: forward + backward of the oracle's tower (torch CPU fp32 autograd over
oracle/clip_ref.py, every visual gradient, no optimiser) on a few frames with the box's host cores.  Synthetic
frames, seeded random weights (no datasets / checkpoints on the box).  The CPU leg is the only place the oracle is
touched."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402


def cpu_baseline(arch, frames=4, repeat=2):
    from eventclip_amd import clip as eclip
    from oracle import clip_ref
    cfg = eclip.arch_config(arch)
    sd = eclip.random_state_dict(cfg, 0)
    leaves = {k: v.float().clone().requires_grad_(True) for k, v in sd.items() if k.startswith('visual.')}
    threads = min(64, os.cpu_count() or 8)
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    imgs = torch.randn(frames, 3, cfg['image_size'], cfg['image_size'])
    best = float('inf')
    for _ in range(repeat):
        for v in leaves.values():
            v.grad = None
        t0 = time.perf_counter()
        feats = clip_ref.encode_image_autograd(leaves, cfg, imgs)
        feats.backward(torch.ones_like(feats))
        best = min(best, time.perf_counter() - t0)
    return dict(frames_per_s=round(frames / best, 2), kind='port', dtype='float32', cores=threads,
                sample='%d frames, forward + backward of every visual parameter, best of %d (%.1f s)' % (frames, repeat, best))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--arch', default='ViT-L/14')
    ap.add_argument('--samples', type=int, default=32)
    ap.add_argument('--views', type=int, default=2)
    ap.add_argument('--classes', type=int, default=1000)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--modes', default='lora,full,bias')
    ap.add_argument('--dtype', default='float16')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--graph', action='store_true', help='replay the step from a hipGraph (recorded after 2 eager steps)')
    a = ap.parse_args()
    from eventclip_amd import _lib, clip as eclip, ft
    from eventclip_amd.clip_cls_ft import FTCLIPClassifier
    dev = _lib.require_gpu()
    B, T, K = a.samples, a.views, a.classes
    cb = None if a.no_cpu_baseline else cpu_baseline(a.arch)
    for mode in a.modes.split(','):
        model = eclip.build_random(a.arch, seed=0, dtype=a.dtype)
        extra = dict(lora='qkvo-16') if mode == 'lora' else (dict(lora=-1, only_bias=True) if mode == 'bias' else dict(lora=-1))
        cd = dict(clip_model=model, prompt='a point cloud image of a {}', class_names=[f'c{i}' for i in range(K)],
                  agg_func='mean', class_tokens=eclip.synthetic_tokens(K), only_conv1=False, only_bias=False,
                  only_ln=False)
        cd.update(extra)
        clf = FTCLIPClassifier(adapter_dict=dict(adapter_type='text-identity', residual=0.95), clip_dict=cd,
                               loss_dict=dict(use_logits_loss=True, use_probs_loss=False)).cuda().train()
        tr = ft.FTTrainer(clf, lr=2e-5, clip_lr=2e-5, total_steps=1000, init_scale=4096.0, graph=a.graph)
        t = tr.tower
        torch.manual_seed(0)
        n = B * T
        patches = (torch.randn(n, t.G, t.kpad, device=dev) * 0.5).to(t.cd)
        patches[:, :, 2 * t.k:] = 0
        valid = torch.ones(B, T, dtype=torch.bool, device=dev)
        labels = torch.randint(0, K, (B,), device=dev)
        row_idx = torch.arange(n, device=dev, dtype=torch.int32).view(B, T)
        data = {'patches': patches, 'row_idx': row_idx, 'valid_mask': valid, 'label': labels}
        for _ in range(max(a.warmup, 4) if a.graph else a.warmup):
            tr.step(data)
        torch.cuda.synchronize()
        _lib.profile_begin()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            loss = tr.step(data)
        tr.resolve()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        prof = _lib.profile_end()
        c = model.cfg
        W, L, S = c['width'], c['layers'], t.S
        M = n * S
        lin = 2.0 * M * 12 * W * W * L                       # the four nn.Linear of every block, forward
        att = 4.0 * S * S * 64 * (W // 64) * n * L
        if mode == 'full':
            flops = 3 * lin + 3.5 * att
        elif mode == 'lora':
            flops = 2 * lin + 2.0 * M * 4 * W * W * L + 3.5 * att      # dX everywhere, dW for q k v o only
        else:
            flops = 2 * lin + 3.5 * att
        ws_gb = t._ws.numel() / 2 ** 30
        print(json.dumps(dict(mode=mode, arch=a.arch, frames_per_step=n, classes=K, dtype=a.dtype,
                              graph=bool(a.graph), ms_per_step=round(dt * 1e3, 2), frames_per_s=round(n / dt, 1),
                              algorithmic_tflop_per_step=round(flops / 1e12, 2),
                              tflops=round(flops / dt / 1e12, 1), trainable_tensors=len(tr.tensors),
                              workspace_gib=round(ws_gb, 2), loss=round(float(loss), 4),
                              loss_scale=tr.scaler.scale, skipped_last=bool(tr.last['skipped']),
                              cpu_baseline=cb,
                              kernel_ms_per_step={p['name']: round(p['total_ms'] / a.steps, 2) for p in prof
                                                  if p['launches']})), flush=True)
        del tr, clf, model
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
