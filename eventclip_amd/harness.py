"""Evaluation loop, meters and data-parallel sharding for the EventCLIP hot path.

* ``evaluate`` reproduces the reference's eval loop (/root/reference/test.py:55-93):
  probs-/logits-based top-1 (and top-5 for N-ImageNet) accumulated in
  ``AverageMeter``s weighted by batch size (nerv.utils.AverageMeter: sum(acc*n)/sum(n)).
* The reference never runs inference on more than one GPU (test.py:36-37,
  use_ddp=False).  Samples are independent, so here a batch is split contiguously
  over the ranks of one node (one process per GPU) and the per-rank logits are
  all-gathered over RCCL/xGMI -- the only collective on the path; a shard is
  <= 2 MB, i.e. latency- not bandwidth-bound.
"""
import torch
import torch.distributed as dist


class AverageMeter:
    """nerv.utils.AverageMeter as test.py uses it: update(val, n) -> avg = sum(val*n)/sum(n)."""

    def __init__(self):
        self.sum, self.count = 0., 0

    def update(self, val, n=1):
        self.sum += float(val) * n
        self.count += n

    @property
    def avg(self):
        return self.sum / max(self.count, 1)


def shard_range(n, rank, world):
    """Contiguous split of n samples: rank r gets [lo, hi); remainders go to the low ranks."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_rows(t, sizes=None):
    """All-gather a [n_local, ...] tensor along dim 0 over the default process group
    (RCCL on GPUs, gloo in the CPU tests).  ``sizes``: per-rank row counts when uneven."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return t
    world = dist.get_world_size()
    if sizes is None or len(set(sizes)) == 1:
        out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t.contiguous())
        return out
    mx = max(sizes)
    pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[:t.shape[0]] = t
    out = torch.empty((world * mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, pad)
    return torch.cat([out[r * mx:r * mx + s] for r, s in enumerate(sizes)], 0)


def gather_out_dict(out_dict, sizes=None, keys=('logits', 'probs')):
    """Gather the per-sample outputs of a sharded forward onto every rank."""
    return {k: all_gather_rows(out_dict[k], sizes) for k in keys}


def batch_accuracies(out_dict, labels, top5=False):
    """The four (or two) accuracies of test.py:62-81 for one batch."""
    probs, logits = out_dict['probs'], out_dict['logits']
    acc = {
        'probs_acc': (probs.argmax(dim=-1) == labels).float().mean().item(),
        'logits_acc': (logits.argmax(dim=-1) == labels).float().mean().item(),
    }
    if top5:
        acc['probs_acc5'] = (probs.topk(5, dim=-1).indices == labels[:, None]).float().sum(-1) \
            .mean().item()
        acc['logits_acc5'] = (logits.topk(5, dim=-1).indices == labels[:, None]).float().sum(-1) \
            .mean().item()
    return acc


@torch.no_grad()
def evaluate(model, batches, is_nin=False, pipeline=None, prefetch=1):
    """test.py:55-93.  ``batches`` yields data_dicts with ``label`` and either the
    reference's ``img``/``valid_mask`` or raw ``events`` (list of arrays) when a
    ``pipeline`` (Event2ImagePipeline) is given; ``prefetch`` batches are then uploaded ahead of the one the
    GPU is working on (0: the synchronous path)."""
    meters = {}
    feeder = None
    if pipeline is not None and prefetch:
        # host-resident event batches: staged through pinned memory and uploaded on a copy stream while the GPU
        # works on the batch before (event2img.HostFeeder) -- the reference's DataLoader prefetch, test.py:36-38.
        # Batches that carry the reference's img / valid_mask instead of events pass through it unchanged
        batches = feeder = pipeline.stream(batches, depth=1 + int(prefetch))
    try:
        for data_dict in batches:
            if pipeline is not None and 'events' in data_dict:
                data_dict = {**pipeline(data_dict['events']), 'label': data_dict['label']}
            data_dict = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in data_dict.items()}
            out_dict = model(data_dict)
            labels = data_dict['label']
            for k, v in batch_accuracies(out_dict, labels, top5=is_nin).items():
                meters.setdefault(k, AverageMeter()).update(v, labels.shape[0])
    finally:
        if feeder is not None:
            feeder.close()       # an exception mid-iteration must not leave the producer thread holding the rings
    return {k: m.avg for k, m in meters.items()}
