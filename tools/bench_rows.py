"""The widened rows of SURVEY.md 8(f) measured on the MI355X, each against its HBM roofline (8 TB/s):
centring (float and packed events), packing, the training augmentation of events, RandAugment on frames, the
TTA view flags of the events kernel, pseudo-label selection.  N-Caltech geometry, the bench's batch (256 samples).

    python tools/bench_rows.py

One JSON line per row: ms per launch, algorithmic bytes, GB/s, fraction of the HBM peak, and `cpu_baseline`: the
oracle's restatement of the reference's CPU code for that row (numpy / C / torch CPU, one process) timed on a bounded
sample of the same input and scaled to the row's full size (`ms_same_work`), so `cpu_baseline.ms_same_work / ms` is
the ratio on this box.  The CPU leg is the only place the oracle is touched."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eventclip_amd import _lib, augment, pseudo_label, randaugment, vis  # noqa: E402
from eventclip_amd.synthetic import make_events  # noqa: E402

PEAK = 8000.0


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def cpu(fn, part, what, repeat=3):
    """Best of `repeat` runs of the oracle on 1 / `part` of the row's input."""
    best = float('inf')
    for _ in range(repeat):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    return dict(ms_same_work=round(best * 1e3 * part, 2), sample=what, sample_ms=round(best * 1e3, 2), cores=1, kind='port')


def line(row, ms, nbytes, cpu_baseline=None, **kw):
    gbps = nbytes / ms / 1e6
    d = dict(row=row, ms=round(ms, 4), algorithmic_bytes=int(nbytes), GBps=round(gbps, 1),
             frac_of_hbm_peak=round(gbps / PEAK, 4), **kw)
    if cpu_baseline:
        cpu_baseline['speedup'] = round(cpu_baseline['ms_same_work'] / ms, 1)
        d['cpu_baseline'] = cpu_baseline
    print(json.dumps(d), flush=True)


def main():
    _lib.require_gpu()
    H, W = 180, 240
    B, n_ev = 256, 200000                      # 256 samples x 10 frames x 20 000 events
    uniq = 16
    ev = np.concatenate([make_events(n_ev, (H, W), seed=i) for i in range(uniq)] * (B // uniq))
    events = torch.from_numpy(ev).cuda()
    offs = torch.arange(B + 1, dtype=torch.int64) * n_ev
    sr = torch.stack([offs[:-1], offs[1:]], 1).cuda()
    n_tot = B * n_ev
    from oracle import event_utils as o_ev, events as o_events, randaugment as o_ra, pseudo_label as o_pl
    one = ev[:n_ev]
    # centring in place: read + write every event
    work = events.clone()
    c_center = cpu(lambda: o_ev.center_events(one.copy(), (H, W)), B, '1 sample of 200 000 events, numpy')
    line('center_events (float32 [n, 4])', timed(lambda: vis.center_events_device(work, sr, (H, W))), 32 * n_tot, events=n_tot,
         cpu_baseline=c_center)
    packed = vis.pack_events_device(events)
    line('pack_events (16 B -> 8 B)', timed(lambda: vis.pack_events_device(events)), 24 * n_tot, events=n_tot,
         cpu_baseline=cpu(lambda: o_ev.packed_fields(one), B, '1 sample of 200 000 events, numpy field extraction'))
    pw = packed.clone()
    line('center_events (packed)', timed(lambda: vis.center_events_device(pw, sr, (H, W))), 16 * n_tot, events=n_tot,
         cpu_baseline=dict(c_center))
    # training augmentation of events: read all, write the survivors (~ all)
    prm = augment.draw_event_augment(B, 20, False, np.random.RandomState(0))
    line('augment_events (shift / flips, drop outside)',
         timed(lambda: augment.augment_events_device(events, [n_ev] * B, prm, (H, W)), n=5), 32 * n_tot, events=n_tot,
         cpu_baseline=cpu(lambda: o_ev.augment_events(one, (7, -5, 1, 1), (H, W)), B, '1 sample of 200 000 events, numpy'))
    # TTA: the four views are flags of the events kernel (same bytes as the plain launch)
    F = B * 10
    fr = torch.tensor([[i * 20000, (i + 1) * 20000] for i in range(F)], dtype=torch.int64).cuda()
    out = torch.empty((F, H, W, 3), dtype=torch.uint8, device='cuda')
    c_e2f = cpu(lambda: o_events.events2frames(one, N=20000, shape=(H, W), grayscale=False), B,
                '1 sample = 10 frames, the C restatement (oracle/events_oracle.c)')
    c_view = cpu(lambda: o_events.events2frames(o_ev.tflip_events(o_ev.hflip_events(one.copy(), (H, W))), N=20000, shape=(H, W),
                                                grayscale=False), B, '1 sample = 10 frames: numpy flips, then the C restatement')
    for name, kw, cb in (('events -> frames', {}, c_e2f),
                         ('events -> frames, flip_x + negate_p view', dict(flip_x=True, negate_p=True), c_view)):
        line(name, timed(lambda: vis.events_to_frames_device(events, fr, (H, W), grayscale=False, out=out,
                                                             max_frame_events=20000, **kw)), F * (16 * 20000 + 3 * H * W), frames=F,
             cpu_baseline=cb)
    # the other two dataset geometries (and the packed 8-byte events of all three)
    pk = vis.pack_events_device(events)
    line('events -> frames, packed events', timed(lambda: vis.events_to_frames_device(pk, fr, (H, W), grayscale=False, out=out,
                                                                                     max_frame_events=20000)),
         F * (8 * 20000 + 3 * H * W), frames=F)
    for gname, gshape, gn, gframes in (('N-Cars 100x120, 12 500 events', (100, 120), 12500, 2560),
                                       ('N-ImageNet 480x640, 70 000 events', (480, 640), 70000, 512),
                                       ('N-ImageNet 480x640, 70 000 events', (480, 640), 70000, 2560)):
        gev = np.concatenate([make_events(gn, gshape, seed=i) for i in range(8)] * (gframes // 8))
        gfr = torch.tensor([[i * gn, (i + 1) * gn] for i in range(gframes)], dtype=torch.int64).cuda()
        gout = torch.empty((gframes, *gshape, 3), dtype=torch.uint8, device='cuda')
        for packed in (False, True):
            ge = torch.from_numpy(vis.pack_events(gev).view(np.int64) if packed else gev).cuda()
            line(f'events -> frames, {gname}{", packed" if packed else ""}',
                 timed(lambda: vis.events_to_frames_device(ge, gfr, gshape, grayscale=False, out=gout, max_frame_events=gn)),
                 gframes * ((8 if packed else 16) * gn + 3 * gshape[0] * gshape[1]), frames=gframes)
            del ge
        del gev, gfr, gout
    # RandAugment: two operators per frame, the same pair for the 10 views of a sample
    frames = out.clone()
    for pair in ((('Rotate', 17.6), ('Contrast', 0.34)), (('ShearX', 0.2), ('Equalize', 0.0)), (('TranslateY', 40.0), ('Sharpness', 0.5)),
                 (('Posterize', 5.0), ('Solarize', 120.0))):
        ops = [list(pair)] * F
        two = frames[:2].cpu().numpy()
        line('randaugment ' + ' + '.join(p[0] for p in pair), timed(lambda: randaugment.apply_ops(frames, ops, (255, 255, 255)), n=5),
             2 * 2 * frames.numel(), frames=F,
             cpu_baseline=cpu(lambda: o_ra.randaugment(two, list(pair), (255, 255, 255)), F // 2, '2 frames, numpy restatement of Pillow', 1))
    # pseudo-label selection: 4 TTA views, N-ImageNet's 1000 classes
    Bp, K = 4096, 1000
    probs = torch.softmax(torch.randn(Bp * 4, K, device='cuda') * 3, -1)
    probs_cpu = probs.cpu()
    line('pseudo_label select (4 views, K = 1000)', timed(lambda: pseudo_label.select(probs, 0.5, tta=True, tta_consistent=True,
                                                                                    tta_min_prob=True)), probs.numel() * 4 + Bp * K * 4, samples=Bp,
         cpu_baseline=cpu(lambda: o_pl.select(probs_cpu, 0.5, tta=True, tta_consistent=True, tta_min_prob=True), 1,
                          'the whole input, torch CPU (its default threads)'))


if __name__ == '__main__':
    main()
