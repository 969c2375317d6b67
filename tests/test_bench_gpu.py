"""bench.py's contract on the GPU box: one JSON line with the required fields, for one rank and for
two ranks (the N > 1 code path; two gloo ranks sharing the single GPU, a functional check)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {'metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better',
            'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline'}
SMALL = ['--steps', '2', '--warmup', '1', '--batch', '4', '--arch', 'ViT-B/32', '--classes', '11',
         '--cpu-baseline-samples', '1', '--cpu-baseline-frames', '2']


def last_json(text):
    lines = [ln for ln in text.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, text[-2000:]
    return json.loads(lines[0])


def test_single_rank_line(hip):
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '1', '--other-configs-batch', '2'] + SMALL, cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    assert REQUIRED <= set(d) and d['n_gpus'] == 1 and d['steps'] == 2 and d['warmup'] == 1
    assert d['value'] > 0 and d['scaling'] == 'weak' and d['vs_baseline'] is None and d['data'] == 'synthetic'
    assert abs(d['value'] - 4 * 10 * 1e3 / d['ms_per_step']) < 1e-6 * d['value']
    roof = d['roofline']
    assert {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'} <= set(roof)
    assert abs(roof['frac'] - roof['achieved'] / roof['peak']) < 1e-9
    cb = d['cpu_baseline']
    assert cb['kind'] == 'port' and cb['value'] > 0 and cb['cores'] >= 1 and cb['unit'] == 'frames/s'
    assert cb['cpu_model'] and cb['event2img_frames_per_s_1proc'] > 0
    assert any(k.startswith('event2img_frames_per_s_pool') for k in cb), cb
    # the events kernel's HBM roofline, with the 16 B per event it reads counted
    re_ = d['roofline_events']
    frames = 4 * 10
    assert re_['algorithmic_bytes_per_launch'] == frames * (16 * 20000 + 3 * 180 * 240)
    assert abs(re_['frac'] - re_['achieved_GBps'] / re_['peak_GBps']) < 1e-9
    assert d['config']['unique_samples'] == 4
    # the price of the 1e-3 mode, measured behind the timed region on the same batch: never part of `value`
    from eventclip_amd import clip as eclip
    tm = d['tolerance_mode']
    assert [tm['precise_blocks'], tm['precise_attn_blocks']] == list(eclip.tolerance_mode_kwargs('ViT-B/32').values())
    assert 'configs_within_1e3' in tm and len(tm['lines']) == 2
    for ln in tm['lines']:
        assert ln['weights'] in ('as_run', 'rounded_to_16_bit') and ln['ms_per_step'] > 0
        assert abs(ln['ratio_to_default'] - ln['ms_per_step'] / ln['default_ms_per_step_interleaved']) < 1e-9
        assert ln['ratio_to_default'] > 1.0          # it is a mode one pays for
    # round 6: the host-fed number, the pipeline default's price and BASELINE configs[2..4] in the same record -- all
    # measured after the timed region, none of them part of `value`
    assert d['value_from_host'] > 0 and d['from_host']['bytes_per_step_per_gpu'] == 4 * 200000 * 16
    assert d['ms_per_step_strict'] > 0
    oc = d['other_configs']
    assert sorted(oc) == ['2', '3', '4']
    for cid, views in (('2', 1), ('3', 2), ('4', 5)):
        ln = oc[cid]
        assert ln['frames_per_step'] == 2 * views and f'configs[{cid}]' in ln['workload']
        assert abs(ln['value'] - ln['frames_per_step'] * 1e3 / ln['ms_per_step']) < 1e-6 * ln['value']
        assert ln['dominant_kernel'] in ln['kernel_ms_per_step'] and 0 <= ln['frac'] < 1      # (two samples: a latency-bound kernel may dominate)
        assert ln['cpu_baseline']['value'] > 0 and ln['cpu_baseline']['kind'] == 'port'


def test_two_ranks_share_the_gpu_over_gloo(hip):
    env = dict(os.environ, EVENTCLIP_DIST_BACKEND='gloo', MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', '29533', 'bench.py', '--gpus', '2'] + SMALL
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    assert d['n_gpus'] == 2 and 'cpu_baseline' not in d
    assert abs(d['value'] - 2 * 4 * 10 * 1e3 / d['ms_per_step']) < 1e-6 * d['value']   # whole-job aggregate
    assert d['config']['parallelism'].startswith('dp2')
    assert d['config']['collective_ranks'] == 2 and len(d['config']['rank_devices']) == 2


def test_gpus_2_launches_itself(hip):
    """`python bench.py --gpus 2` the way the driver runs `--gpus 1`: no launcher around it.  The parent
    starts the ranks as a child torch.distributed.run and relays its exit code."""
    env = dict(os.environ, EVENTCLIP_DIST_BACKEND='gloo')
    env.pop('WORLD_SIZE', None)
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '2'] + SMALL, cwd=ROOT, capture_output=True,
                       text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    assert d['n_gpus'] == 2 and d['config']['collective_ranks'] == 2
    # a failing child is reported, not swallowed
    bad = subprocess.run([sys.executable, 'bench.py', '--gpus', '2', '--arch', 'no-such-arch'] + SMALL[:4],
                         cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert bad.returncode != 0


@pytest.mark.parametrize('share', [False, True])
def test_rccl_that_cannot_come_up_fails_fast_and_readably(hip, share):
    """`--gpus 2` over nccl (= RCCL) on the ONE GPU of this box cannot work: without a device of its own a rank stops
    before it touches the library; with both ranks forced onto device 0 (EVENTCLIP_DIST_SHARE_DEVICE=1) RCCL itself
    refuses or never answers, and the short init timeout / the first 4-byte all-gather's watchdog end the run.  Either
    way: non-zero exit in well under the driver's limit, and a diagnostic block that names rank, device and the HSA_* /
    NCCL_* environment -- the failure mode of a fabric that does not come up on a real 8-GPU node."""
    import time
    import torch
    if torch.cuda.device_count() != 1:
        pytest.skip('needs a single-GPU box')
    env = dict(os.environ, EVENTCLIP_DIST_TIMEOUT='40')
    env.pop('WORLD_SIZE', None)
    env.pop('EVENTCLIP_DIST_BACKEND', None)
    if share:
        env['EVENTCLIP_DIST_SHARE_DEVICE'] = '1'
    t0 = time.time()
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '2', '--no-cpu-baseline'] + SMALL, cwd=ROOT, capture_output=True,
                       text=True, timeout=400, env=env)
    assert r.returncode != 0 and time.time() - t0 < 300
    assert '[bench] rank' in r.stderr and 'env:' in r.stderr and 'HSA_ENABLE_IPC_MODE_LEGACY' in r.stderr, r.stderr[-3000:]
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith('{')]      # no line is printed for a run that did not happen


@pytest.mark.parametrize('config,batch,expect', [
    (2, 3, dict(scaling='weak', samples=[3, 3], views=1)),            # per-GPU batch, one short view per sample
    (3, 5, dict(scaling='strong', samples=[3, 2], views=2)),          # GLOBAL batch over the ranks, uneven shards
    (4, 3, dict(scaling='strong', samples=[2, 1], views=5)),
])
def test_other_baseline_configs_two_ranks_over_gloo(hip, config, batch, expect):
    """bench.py --config 2 | 3 | 4 (BASELINE.json configs[2..4]) on two gloo ranks sharing the GPU: the geometry,
    views and adapter of the config, weak (per-GPU batch) or strong (global batch split by harness.shard_range)
    scaling, the whole-job value, every rank's own ms per step and the all-gather's time in the line."""
    env = dict(os.environ, EVENTCLIP_DIST_BACKEND='gloo')
    env.pop('WORLD_SIZE', None)
    cmd = [sys.executable, 'bench.py', '--gpus', '2', '--config', str(config), '--batch', str(batch), '--steps', '2',
           '--warmup', '1', '--arch', 'ViT-B/32', '--classes', '11', '--no-cpu-baseline']
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    assert REQUIRED <= set(d) and d['n_gpus'] == 2 and d['scaling'] == expect['scaling']
    cfg = d['config']
    assert f'configs[{config}]' in cfg['workload'] and cfg['samples_per_rank'] == expect['samples']
    frames = sum(expect['samples']) * expect['views']
    assert cfg['frames_per_step_total'] == frames
    assert abs(d['value'] - frames * 1e3 / d['ms_per_step']) < 1e-6 * d['value']      # whole-job aggregate
    assert len(d['ms_per_step_per_rank']) == 2 and all(0 < t <= d['ms_per_step'] * 1.001 for t in d['ms_per_step_per_rank'])
    assert d['all_gather_ms_per_step'] is not None and d['all_gather_ms_per_step'] >= 0
    assert (cfg['adapter'] is None) == (config == 3)


@pytest.mark.parametrize('config,batch,expect', [
    (3, 20, dict(scaling='strong', samples=[3, 3, 3, 3, 2, 2, 2, 2], views=2)),    # uneven shards over 8 ranks
    (1, 1, dict(scaling='weak', samples=[1] * 8, views=10)),
])
def test_eight_ranks_over_gloo(hip, config, batch, expect):
    """The rank count the scaling run uses (/root/reference/scripts/sbatch_run.sh:46-51 launches
    `--nproc_per_node=$GPUS`), exercised before the driver does: eight gloo ranks sharing the one GPU.  Uneven
    shards (the all-gather's padding path), eight entries in rank_devices / ms_per_step_per_rank, the whole-job
    value.  A functional check of the N = 8 code path, not a measurement."""
    env = dict(os.environ, EVENTCLIP_DIST_BACKEND='gloo')
    env.pop('WORLD_SIZE', None)
    cmd = [sys.executable, 'bench.py', '--gpus', '8', '--config', str(config), '--batch', str(batch), '--steps', '2',
           '--warmup', '1', '--arch', 'ViT-B/32', '--classes', '11', '--no-cpu-baseline']
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = last_json(r.stdout)
    assert REQUIRED <= set(d) and d['n_gpus'] == 8 and d['scaling'] == expect['scaling']
    cfg = d['config']
    assert cfg['collective_ranks'] == 8 and len(cfg['rank_devices']) == 8
    assert cfg['samples_per_rank'] == expect['samples']
    frames = sum(expect['samples']) * expect['views']
    assert cfg['frames_per_step_total'] == frames
    assert abs(d['value'] - frames * 1e3 / d['ms_per_step']) < 1e-6 * d['value']
    assert len(d['ms_per_step_per_rank']) == 8 and all(0 < t <= d['ms_per_step'] * 1.001 for t in d['ms_per_step_per_rank'])
    assert d['all_gather_ms_per_step'] is not None and d['all_gather_ms_per_step'] >= 0


def test_config1_line_keeps_its_keys(hip):
    """--config 1 is the default line: same metric string and config keys as before the other configs existed."""
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '1', '--config', '1', '--no-cpu-baseline', '--no-dvfs', '--no-other-configs', '--no-from-host', '--no-strict-line'] + SMALL,
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    assert d['metric'] == 'event-frames/sec (whole node) ViT-L/14 zero-shot @224' and d['scaling'] == 'weak'
    assert 'samples_per_rank' not in d['config'] and 'ms_per_step_per_rank' not in d
    assert d['config']['workload'].endswith('batch=4 samples x 10 views per GPU (configs[1])')


@pytest.mark.parametrize('flags,tag', [(['--precise-blocks', '2'], 'first 2 blocks'), (['--precise-blocks', '2', '--f16-weights'], 'first 2 blocks'),
                                       (['--precise'], 'split-precision image tower')])
def test_split_precision_lines_are_labelled(hip, flags, tag):
    """--precise-blocks N / --precise / --f16-weights: lines of their own -- the metric string says which mode, the
    config says how the weights were made -- never to be mistaken for the headline."""
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '1', '--no-cpu-baseline', '--no-dvfs', '--no-from-host', '--no-strict-line'] + flags + SMALL,
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    assert d['metric'].startswith('event-frames/sec (whole node) ViT-L/14 zero-shot @224 -- ') and tag in d['metric']
    assert ('rounded to 16 bit first' in d['config']['weights']) == ('--f16-weights' in flags)
    assert d['value'] > 0
