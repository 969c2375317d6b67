"""Two-rank check of FTTrainer's data-parallel step (run under torch.distributed.run; tests/test_ft_train_gpu.py).

Every rank builds the same seeded tiny tower, takes its contiguous half of a seeded batch, and steps; rank 0
also steps a second trainer on the WHOLE batch without a process group in the way.  The averaged gradients of
the halves are the gradient of the whole batch's mean loss, so after the step the parameters of the two must
agree -- compared here on the gradients the optimiser receives.  Prints one JSON line from rank 0."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def build(mode):
    from eventclip_amd import clip as eclip, ft
    from eventclip_amd.clip_cls_ft import FTCLIPClassifier
    cfg = dict(image_size=48, patch=16, width=128, layers=2, embed_dim=32, text_width=64, text_heads=1, text_layers=1,
               context_length=77, vocab_size=128)
    model = eclip.CLIP(cfg, eclip.random_state_dict(cfg, seed=0), full_last_block=True).cuda()
    K = 7
    cd = dict(clip_model=model, prompt='a point cloud image of a {}', class_names=[f'c{i}' for i in range(K)],
              agg_func='mean', class_tokens=eclip.synthetic_tokens(K), only_conv1=False, only_bias=False, only_ln=False,
              lora='qkvo-4' if mode == 'lora' else -1)
    clf = FTCLIPClassifier(adapter_dict=dict(adapter_type='text-identity', residual=True), clip_dict=cd,
                           loss_dict=dict(use_logits_loss=True, use_probs_loss=False)).cuda().train()
    # the LoRA factors' initial values: the same seed on every rank -- or, with FT_DDP_SEED_PER_RANK=1, a
    # different one per rank (the usual per-rank seed offset): FTTrainer then has to start everyone from rank 0's
    per_rank = os.environ.get('FT_DDP_SEED_PER_RANK') == '1' and dist.is_initialized()
    torch.manual_seed(5 + (dist.get_rank() if per_rank else 0))
    tr = ft.FTTrainer(clf, lr=1e-2, clip_lr=1e-3, total_steps=100, warmup_steps_pct=0.0, init_scale=256.0)
    if tr.lora:
        for k, p in tr.lora.params.items():
            if 'lora_up' in k:
                p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(len(k))).cuda() * 0.05)
        tr.lora.merge()
    return clf, tr


def batch(B):
    g = torch.Generator().manual_seed(11)
    imgs = torch.randn(B, 2, 3, 48, 48, generator=g)
    valid = torch.ones(B, 2, dtype=torch.bool)
    labels = torch.randint(0, 7, (B,), generator=g)
    return imgs, valid, labels


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else 'full'
    dist.init_process_group(os.environ.get('EVENTCLIP_DIST_BACKEND', 'nccl'))
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)) % max(torch.cuda.device_count(), 1))
    B = 4 * world
    imgs, valid, labels = batch(B)
    lo, hi = rank * 4, rank * 4 + 4
    clf, tr = build(mode)

    def same_everywhere():
        ok = True
        for k in sorted(tr.tensors):
            both = [torch.empty_like(tr.tensors[k]) for _ in range(world)]
            dist.all_gather(both, tr.tensors[k].contiguous())
            ok = ok and all(torch.equal(both[0], b) for b in both[1:])
        return ok
    equal_start = same_everywhere()
    loss = tr.step({'img': imgs[lo:hi].cuda(), 'valid_mask': valid[lo:hi].cuda(), 'label': labels[lo:hi].cuda()})
    tr.resolve()
    losses = [torch.zeros(1, device='cuda') for _ in range(world)]
    dist.all_gather(losses, loss.reshape(1))
    mine = {k: v.clone() for k, v in tr.last['grads'].items()}
    out = None
    if rank == 0:
        import eventclip_amd.ft as ftmod
        saved = ftmod.dist
        class _NoDist:                         # the single-process reference run must not see the process group
            @staticmethod
            def is_available():
                return False
        ftmod.dist = _NoDist
        clf1, tr1 = build(mode)
        whole = tr1.step({'img': imgs.cuda(), 'valid_mask': valid.cuda(), 'label': labels.cuda()})
        tr1.resolve()
        ftmod.dist = saved
        worst = 0.0
        for k, v in tr1.last['grads'].items():
            if k.endswith('attn.in_proj_bias'):
                continue                       # its key third is zero in exact arithmetic: rounding noise only
            worst = max(worst, ((v - mine[k]).norm() / v.norm().clamp_min(1e-30)).item())
        out = dict(seed_per_rank=os.environ.get('FT_DDP_SEED_PER_RANK') == '1', params_equal_at_start=equal_start,
                   mode=mode, world=world, loss_mean_of_ranks=float(torch.cat(losses).mean()), loss_whole=float(whole),
                   worst_grad_rel_l2=worst, tensors=len(mine), skipped=bool(tr.last['skipped']))
    for _ in range(2):
        tr.step({'img': imgs[lo:hi].cuda(), 'valid_mask': valid[lo:hi].cuda(), 'label': labels[lo:hi].cuda()})
    tr.resolve()
    equal_end = same_everywhere()
    dist.barrier()
    if rank == 0:
        out['params_equal_after_steps'] = equal_end
        print(json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
