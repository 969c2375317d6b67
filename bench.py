"""Headline benchmark: event-frames/sec through the whole hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one batch of the BASELINE.json config "N-Caltech101 zero-shot, ViT-L/14,
RGB-polarity event2img, batch=256" (SURVEY.md 8(d) C2): raw events already resident
in HBM -> histogram frames -> CLIP preprocess -> ViT-L/14 image tower -> logits against
the cached text features -> per-sample aggregation (-> RCCL all-gather of the logits
when N > 1).  Every sample has 10 x 20000 events, i.e. 10 valid views, so a step
pushes 2560 frames per GPU; weak scaling (each rank gets its own 256 samples).
Weights are seeded random (no checkpoints ship), data is synthetic.

Rank 0 prints one JSON line; `roofline` comes from HIP events the library records
around its own launches during the timed steps, `cpu_baseline` is the CPU oracle
(test infrastructure) timed on a bounded sample on the host.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_TFLOPS = 2500.0   # dense bf16/f16 MFMA, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0       # HBM3E spec, MI355X_MICROARCH.md
MFMA_KERNELS = ('gemm_kernel', 'attention_kernel')


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=256, help='samples per GPU per step')
    ap.add_argument('--arch', default='ViT-L/14')
    ap.add_argument('--dtype', default='float16', choices=['float16', 'bfloat16'])
    ap.add_argument('--chunk', type=int, default=2560, help='frames per pass through the tower')
    ap.add_argument('--classes', type=int, default=101)
    ap.add_argument('--cpu-baseline-samples', type=int, default=3)
    ap.add_argument('--cpu-baseline-frames', type=int, default=10, help='views per baseline sample')
    ap.add_argument('--packed-events', action='store_true',
                    help='feed the 8-byte packed event form (SURVEY 8(f)) instead of float32 [n, 4]')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-dvfs', action='store_true', help='skip the clock / power sampling steps')
    return ap.parse_args()


def cpu_baseline(cfg, sd, tokens, events, quantize_args, n_samples, max_frames):
    """The CPU oracle chain on a bounded sample of the same workload (rank 0, N=1 only)."""
    from oracle import classify as oc
    from oracle import clip_ref
    from oracle import events as oe
    from oracle import preprocess as op
    threads = min(os.cpu_count() or 1, 64)   # more threads than this slow torch's CPU GEMMs down
    torch.set_num_threads(threads)
    qa = {k: v for k, v in quantize_args.items()
          if k not in ('max_imgs', 'split_method', 'convert_method')}
    text = torch.nn.functional.normalize(clip_ref.encode_text(sd, cfg, tokens), dim=-1)  # cached
    t0 = time.perf_counter()
    n_frames = 0
    for ev in events[:n_samples]:
        ev = ev[:max_frames * qa['N']]
        frames = oe.events2frames(ev, 'event_count', 'event_histogram', shape=(180, 240), **qa)
        imgs = torch.from_numpy(op.preprocess(frames, cfg['image_size']))
        feats = clip_ref.encode_image(sd, cfg, imgs)
        valid = torch.ones(1, frames.shape[0], dtype=torch.bool)
        oc.zs_forward(feats, valid, text, 100.0, 'mean')
        n_frames += frames.shape[0]
    dt = time.perf_counter() - t0
    return {'value': n_frames / dt, 'unit': 'frames/s', 'cores': threads, 'kind': 'port',
            'sample': f'{n_samples} sample(s) cut to {n_frames} frames of the same workload through the '
                      f'CPU oracle (C events2frames + numpy Pillow-bicubic + torch fp32 '
                      f'{threads}-thread ViT), {dt:.1f} s'}


def sample_dvfs(step, fence, n_steps=6):
    """Shader clock and socket power while the step runs (untimed extra steps, after the timed
    region): `rocm-smi` is polled from a thread that never touches the HIP context.  The MFMA peak
    in `roofline` is quoted at the 2.4 GHz boost clock; under its 1400 W cap the chip sustains less
    on this workload, and `peak_at_sclk` restates the peak at the clock that was actually observed."""
    import re
    import subprocess
    import threading
    samples, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            try:
                out = subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True,
                                     text=True, timeout=20).stdout
            except Exception:
                return
            clk = re.search(r'GPU\[0\].*sclk clock level.*\((\d+)Mhz\)', out)
            pw = re.search(r'GPU\[0\].*Power \(W\): ([\d.]+)', out)
            if clk and pw:
                samples.append((int(clk.group(1)), float(pw.group(1))))

    th = threading.Thread(target=poll, daemon=True)
    th.start()
    for _ in range(n_steps):
        step()
    fence()
    stop.set()
    th.join(timeout=30)
    busy = [s for s in samples if s[1] > 600.]      # samples taken while the GPU was loaded
    if not busy:
        return None
    sclk = sum(s[0] for s in busy) / len(busy)
    return {'sclk_mhz': sclk, 'socket_power_w': sum(s[1] for s in busy) / len(busy),
            'samples': len(busy), 'boost_mhz': 2400,
            'peak_at_sclk': PEAK_MFMA_TFLOPS * sclk / 2400.}


def main():
    a = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    # one process per GPU over RCCL; EVENTCLIP_DIST_BACKEND=gloo lets two ranks share one GPU so the
    # N > 1 code path can be exercised on a single-GPU box (a functional check, not a measurement)
    backend = os.environ.get('EVENTCLIP_DIST_BACKEND', 'nccl')
    local = local % max(torch.cuda.device_count(), 1) if backend != 'nccl' else local
    torch.cuda.set_device(local)
    if world > 1:
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend)
    assert world == a.gpus, f'--gpus {a.gpus} but WORLD_SIZE={world}'

    from eventclip_amd import _lib
    from eventclip_amd import clip as eclip
    from eventclip_amd.clip_cls import ZSCLIPClassifier
    from eventclip_amd.event2img import Event2ImagePipeline
    from eventclip_amd.harness import all_gather_rows
    from eventclip_amd.synthetic import GEOMETRY, make_events

    geo = GEOMETRY['n_caltech']
    T, N = 10, geo['N']
    quantize_args = dict(max_imgs=T, N=N, split_method='event_count',
                         convert_method='event_histogram', grayscale=False,
                         count_non_zero=geo['count_non_zero'],
                         background_mask=geo['background_mask'])

    # ---- model: seeded random ViT-L/14 CLIP, text features cached once ----
    cfg = eclip.arch_config(a.arch)
    sd = eclip.random_state_dict(cfg, seed=2)
    clip_model = eclip.CLIP(cfg, sd, dtype=a.dtype, chunk=a.chunk).cuda().eval()
    tokens = eclip.synthetic_tokens(a.classes, seed=2)
    model = ZSCLIPClassifier(clip_dict=dict(
        clip_model=clip_model, prompt='a point cloud image of a {}',
        class_names=[f'class {i}' for i in range(a.classes)], agg_func='mean',
        class_tokens=tokens)).cuda().eval()
    model.get_text_feats()

    # ---- data: per-rank batch of event streams, resident in HBM ----
    uniq = min(a.batch, 32)
    evs = [make_events(T * N, geo['resolution'], seed=2 * 100003 + rank * 1000 + i)
           for i in range(uniq)]
    n_events = [T * N] * a.batch
    events = torch.from_numpy(np.concatenate([evs[i % uniq] for i in range(a.batch)])).cuda()
    if a.packed_events:
        from eventclip_amd.vis import pack_events_device
        events = pack_events_device(events)
    pipe = Event2ImagePipeline(geo['resolution'], geo['max_n'], quantize_args,
                               n_px=cfg['image_size'], patch=cfg['patch'], kpad=clip_model.kpad,
                               dtype=clip_model.compute_dtype)
    pipe.strict = False   # no host sync inside the step (bounds are checked by the tests)
    frames_per_step = a.batch * T

    def step():
        batch = pipe(events, n_events)
        out = model(batch)
        if world > 1:
            out['logits'] = all_gather_rows(out['logits'])
        return out

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    _lib.profile_begin()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    fence()
    dt = time.perf_counter() - t0
    prof = _lib.profile_end()
    assert int(out['valid_masks'].sum()) == frames_per_step
    if world > 1:
        t = torch.tensor([dt], device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        value = world * frames_per_step * a.steps / dt
        # ---- roofline of the dominant kernel, from the live HIP-event records ----
        dom = max(prof, key=lambda e: e['total_ms'])
        avg_ms = dom['total_ms'] / dom['launches']
        if dom['name'].startswith(MFMA_KERNELS):
            achieved = dom['flops'] / dom['launches'] / (avg_ms * 1e-3) / 1e12
            roof = {'bound': 'mfma', 'achieved': achieved, 'peak': PEAK_MFMA_TFLOPS,
                    'unit': 'TFLOP/s', 'frac': achieved / PEAK_MFMA_TFLOPS}
        else:
            achieved = dom['bytes'] / dom['launches'] / (avg_ms * 1e-3) / 1e9
            roof = {'bound': 'hbm', 'achieved': achieved, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                    'frac': achieved / PEAK_HBM_GBS}
        roof.update(kernel=dom['name'], launches_per_step=dom['launches'] / a.steps,
                    avg_launch_ms=avg_ms, traffic=None)
        traffic_file = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(traffic_file):
            tr = json.load(open(traffic_file))
            if tr.get('kernel') == dom['name']:
                roof['traffic'] = tr.get('hbm_bytes_per_launch')
        gpu_ms = sum(e['total_ms'] for e in prof) / a.steps
        breakdown = {e['name']: round(e['total_ms'] / a.steps, 3) for e in prof}
        res = {
            'metric': 'event-frames/sec (whole node) ViT-L/14 zero-shot @224',
            'value': value, 'unit': 'frames/s', 'n_gpus': world, 'steps': a.steps,
            'warmup': a.warmup, 'ms_per_step': dt / a.steps * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f16' if a.dtype == 'float16' else 'bf16',
            'data': 'synthetic',
            'config': {'workload': f'N-Caltech101 zero-shot, {a.arch}, RGB-polarity event2img, '
                                   f'batch={a.batch} samples x {T} views per GPU (configs[1])',
                       'frames_per_step_per_gpu': frames_per_step, 'classes': a.classes,
                       'events_per_frame': N, 'resolution': list(geo['resolution']),
                       'event_format': 'packed 8 B' if a.packed_events else 'float32 [n, 4]',
                       'tower_chunk_frames': a.chunk, 'weights': 'seeded random',
                       'last_block': ('every token' if clip_model.full_last_block else
                                      'keys/values for every token; query, out_proj, MLP for the class '
                                      'token only (bit-identical encode_image output)'),
                       'parallelism': f'dp{world}, all-gather of logits' if world > 1 else 'single GPU'},
            'roofline': roof,
            'kernel_ms_per_step': breakdown, 'kernel_ms_per_step_total': gpu_ms,
            'kernel_launches_per_step': {e['name']: e['launches'] / a.steps for e in prof},
            'kernel_algorithmic_bytes_per_launch': {e['name']: e['bytes'] / e['launches'] for e in prof},
        }
        if world == 1 and not a.no_dvfs:
            dv = sample_dvfs(step, fence)
            if dv:
                res['dvfs'] = dv
                if roof['bound'] == 'mfma':
                    roof['frac_of_peak_at_sclk'] = roof['achieved'] / dv['peak_at_sclk']
        if world == 1 and not a.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline(cfg, sd, tokens, evs, quantize_args,
                                               a.cpu_baseline_samples, a.cpu_baseline_frames)
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
