"""BASELINE.json configs as end-to-end parity cases: events -> frames -> preprocess -> CLIP tower ->
(adapter) -> logits on the MI355X against the oracle chain, at the FULL depth of the named
architecture (24 / 12 vision blocks) and a batch small enough for the fp32 CPU oracle to finish in
seconds.  Every config runs twice:

  'init'    weights at OpenAI's init scales: north_star's tolerance -- 1e-3 relative to max |logit| -- is asserted
            on full_logits and on the aggregated logits.  On these weights ~98 % of the feature vector is the same
            whatever the input, so that number mostly measures a constant;
  'signal'  weights and frames whose image features are input-dependent (>= 30 % of their norm, asserted:
            clip.random_state_dict(qk_gain, branch_gain), events concentrated in a blob): the error is ALSO
            measured against the input-dependent part of the logits (centred error), the yardstick is the
            reference's own GPU arithmetic (fp16 weights and activations, oracle/clip_ref.py
            emulate='fp16_reference') -- the HIP path's error against the fp32 oracle must not exceed it on
            either metric --, and top-1 / top-5 must agree with the fp32 oracle.  With 16-bit GEMM operands the
            max-normalised error is 1.5e-3 .. 2e-3 there (tools/rounding_budget.py --qk 2.5 --branch 4 --blob 0.7:
            nine rounding groups of 3e-4 .. 6e-4 each), the reference's fp16 path 2.4e-3 .. 3.2e-3.

The full batch sizes run through size-independent properties (test_config*_full_size_properties)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


from config_cases import (CASES, SIGNAL_GAINS, build_inputs, load_golden, fingerprint, make_events_batch,   # noqa: E402,F401
                          make_weights, oracle_case, oracle_forward, quantize_args)

LOGIT_TOL = 1e-3   # north_star: logits within 1e-3 relative of the reference's (fp32 oracle)
# Input-dependent weights: the max-normalised error of full_logits against the fp32 oracle, per config, pinned at
# 1.5 x what the HIP path measures on draw 0 (profiles/r5_parity.txt: 1.3e-3 / 1.9e-3 / 1.3e-3 / 4.9e-3 / 5.3e-3; all eight draws:
# profiles/r6_parity_seeds.txt, worst 2.4e-3 / 1.9e-3 / 2.5e-3 / 6.7e-3 / 7.2e-3; the two N-ImageNet
# cases were 3.1e-3 / 2.2e-3 on round 4's wide blobs, where 8 - 15 % of their features depended on the input: they now
# carry 25 - 27 %, make_events_batch) -- a regression of 2 x fails.  north_star's 1e-3 is NOT met there with 16-bit GEMM
# operands: the tolerance mode (ec_vit_weights.precise_blocks) gets every config under it over the draws of
# tests/config_cases.py (test_tolerance_mode_meets_1e3_over_draws); DESIGN.md 3.3 and the header say so.
SIGNAL_TOL = {'n_caltech/ViT-B/32': 2.0e-3, 'n_caltech/ViT-L/14': 2.9e-3, 'n_cars/ViT-L/14': 2.0e-3,
              'n_imagenet/ViT-L/14@336px': 7.4e-3, 'n_imagenet/ViT-L/14': 8.0e-3}
WEIGHTS = pytest.mark.parametrize('weights', ['init', 'signal'])
LINE_TAG = ''      # appended to the config's name in the printed / recorded parity lines (the precise_blocks runs)
# EC_LIVE_ORACLE=1: ignore tests/golden/configs_oracle_*.npz and recompute every oracle chain on this host
LIVE_ORACLE = os.environ.get('EC_LIVE_ORACLE', '0') not in ('', '0')


def hip_case(inp, **clip_kw):
    """The HIP path on a case of config_cases.build_inputs: (out_dict, classifier, pipeline)."""
    import torch
    from eventclip_amd import clip as eclip
    from eventclip_amd.clip_cls import FSCLIPClassifier, ZSCLIPClassifier
    from eventclip_amd.event2img import Event2ImagePipeline
    case, g = inp['case'], inp['g']
    m = eclip.CLIP(inp['cfg'], inp['sd'], **clip_kw).cuda().eval()
    clip_dict = dict(clip_model=m, prompt='a point cloud image of a {}', class_names=[str(i) for i in range(case['K'])],
                     agg_func='mean', class_tokens=inp['tokens'])
    if case['kind'] == 'zs':
        model = ZSCLIPClassifier(clip_dict=clip_dict).cuda().eval()
    else:
        model = FSCLIPClassifier(
            adapter_dict=dict(adapter_type='text-trans', in_dim=768, d_model=256, num_heads=4, ffn_dim=1024,
                              norm_first=True, num_layers=2, residual=case['residual']),
            clip_dict=clip_dict, loss_dict=dict(use_logits_loss=True, use_probs_loss=False))
        model.adapter.load_state_dict(inp['adapter_sd'])
        model = model.cuda().eval()
    pipe = Event2ImagePipeline(g['resolution'], case['max_n'] or g['max_n'], inp['qa'], n_px=case['n_px'],
                               patch=case['patch'], kpad=m.kpad)
    with torch.no_grad():
        out = model(pipe(inp['evs']))
    return out, model, pipe


def oracle_for(inp, live=False, need_emu=True):
    """(want out_dict, image features, fp16-reference emulation out_dict or None, 'shipped' | 'live').  The fp32 oracle
    chain of a full-depth ViT-L/14 costs tens of seconds of host time per case: its outputs are computed once in the
    build container (tools/make_golden_configs.py) and shipped; they are used when the regenerated inputs carry the
    fingerprint the file records, else (or with live=True / EC_LIVE_ORACLE=1) the chain runs here."""
    import torch
    gold = None if (live or LIVE_ORACLE) else load_golden(inp['c'], inp['weights'], inp['draw'])
    if gold is not None and not np.allclose(gold['fingerprint'], fingerprint(inp), rtol=1e-9, atol=0):
        import warnings
        warnings.warn(f"configs[{inp['c']}] {inp['weights']} draw {inp['draw']}: this host regenerates different inputs "
                      f"than the build container ({fingerprint(inp)} vs {gold['fingerprint']}); running the oracle live")
        gold = None
    if gold is not None:
        vm = torch.from_numpy(gold['valid_masks'])
        want = dict(full_logits=torch.from_numpy(gold['full_logits']), logits=torch.from_numpy(gold['logits']), valid_masks=vm)
        emu = dict(full_logits=torch.from_numpy(gold['emu_full_logits']), logits=torch.from_numpy(gold['emu_logits']),
                   valid_masks=vm)
        return want, torch.from_numpy(gold['feats']), emu, 'shipped'
    want, feats = oracle_case(inp)
    emu = oracle_case(inp, emulate='fp16_reference')[0] if need_emu else None
    return want, feats, emu, 'live'


def run_config(c, weights, draw=0, live=False, tol=None, **clip_kw):
    """One case of tests/config_cases.py through the HIP path and against the oracle (shipped outputs of the build
    container's run unless live): on 'signal' weights also against the yardstick.  -> (inputs, out, want, pipe)."""
    inp = build_inputs(c, weights, draw)
    out, _, pipe = hip_case(inp, **clip_kw)
    want, feats, emu, src = oracle_for(inp, live=live)
    key = inp['case']['key']
    name = f'configs[{c}]' + (f' draw {draw}' if draw else '') + (' (live oracle)' if src == 'live' else '')
    if weights.startswith('signal'):
        # the input-dependent share of the features is asserted on the historical draw (the gains were set on it);
        # the other draws are held to 80 % of it (measured 0.17 .. 0.40, tools/make_golden_configs.py's log)
        share = SIGNAL_GAINS[key][2] * (1.0 if draw == 0 else 0.8)
        check(out, want, feats=feats, emu=emu, logit_tol=SIGNAL_TOL[key] if tol is None else tol, name=name, min_share=share)
    else:
        check(out, want)
    return inp, out, want, pipe


def record_parity(line):
    """EC_PARITY_TABLE=<file>: the measured parity lines are appended there (profiles/r4_parity.txt is such a file,
    regenerated whenever a tower kernel changes: tools/profile_round.sh)."""
    import os
    path = os.environ.get('EC_PARITY_TABLE')
    if path:
        with open(path, 'a') as f:
            f.write(line + '\n')


def logit_errors(out, want):
    """{key: (max-normalised, centred)} for full_logits (rows = valid views) and logits (rows = samples).
    max-normalised = max|err| / max|logit| (north_star's 1e-3).  Centred = max|err| / max|logit - its per-class
    mean over the rows|: the error against the part of the logits that depends on the INPUT.  (The reference's
    per-class mean is subtracted from both sides, so a constant per-class offset of the path under test
    still counts as error.)"""
    vm = want['valid_masks']
    res = {}
    for k in ('full_logits', 'logits'):
        g, w = out[k].float(), want[k].float()
        if k == 'full_logits':
            g, w = g[vm], w[vm]
        err = float((g - w).abs().max())
        spread = float((w - w.mean(0)).abs().max()) if w.shape[0] > 1 else float('nan')   # one row: no mean to take
        res[k] = (err / float(want['full_logits'].abs().max()), err / spread if spread == spread else 0.0)
    return res


def signal_share(feats):
    """share of the feature norm that varies with the input"""
    return float((feats - feats.mean(0)).norm() / feats.norm())


def ranks_agree(got, want, k, eps):
    """top-k of `got` against the fp32 oracle's, row by row: the i-th pick of the path under test must be, in the
    ORACLE's values, within eps of the oracle's own i-th best -- orders may differ only between classes the
    oracle itself puts closer together than the measured error."""
    import torch
    k = min(k, want.shape[-1])
    idx = got.topk(k, -1).indices
    best = want.topk(k, -1).values
    return bool((torch.gather(want, -1, idx) >= best - eps).all())


def check(out, want, feats=None, emu=None, logit_tol=LOGIT_TOL, name='', min_share=0.3):
    """north_star's 1e-3 on the max-normalised logits; with the fp16-reference emulation given, also: the features
    the check runs on are input-dependent (>= min_share of their norm), the CENTRED error of the HIP path does
    not exceed the reference's own GPU arithmetic's, and top-1 / top-5 agree with the fp32 oracle."""
    import torch
    assert torch.equal(out['valid_masks'].cpu(), want['valid_masks'])
    o = {k: v.cpu() for k, v in out.items()}
    e = logit_errors(o, want)
    if emu is None:
        for k in ('full_logits', 'logits'):
            assert e[k][0] < logit_tol, (k, e[k])
        return
    share = signal_share(feats)
    ee = logit_errors(emu, want)
    line = (f'[{name}{LINE_TAG}] input-dependent share of the image features {share:.2f}; full_logits error vs the fp32 '
            f'oracle, max-normalised / centred: HIP {e["full_logits"][0]:.2e} / {e["full_logits"][1]:.2e}, '
            f'fp16-reference emulation {ee["full_logits"][0]:.2e} / {ee["full_logits"][1]:.2e}; aggregated logits: '
            f'HIP {e["logits"][0]:.2e} / {e["logits"][1]:.2e}, emulation {ee["logits"][0]:.2e} / {ee["logits"][1]:.2e}'
            f'; bound {logit_tol:.1e}')
    print('\n' + line)
    record_parity(line)
    for k in ('full_logits', 'logits'):
        assert e[k][0] < logit_tol, (k, e[k])
    assert share >= min_share, share
    # the yardstick: strictly on full_logits (every valid view); the aggregated logits of a handful of samples are
    # the same errors averaged over 1 .. T views and their maximum a single draw: 25 % slack there
    assert e['full_logits'][0] <= ee['full_logits'][0] and e['full_logits'][1] <= ee['full_logits'][1], (e, ee)
    assert e['logits'][0] <= 1.25 * ee['logits'][0] and e['logits'][1] <= 1.25 * ee['logits'][1], (e, ee)
    vm = want['valid_masks']
    mag = float(want['full_logits'].abs().max())
    for k, g, w in (('logits', o['logits'], want['logits']), ('full_logits', o['full_logits'][vm], want['full_logits'][vm])):
        assert torch.equal(g.argmax(-1), w.argmax(-1)) or ranks_agree(g, w, 1, 2 * e[k][0] * mag), k
        assert ranks_agree(g, w, 5, 2 * e[k][0] * mag), k


def quantize_args(geo_name, T, grayscale=True):
    from eventclip_amd.synthetic import GEOMETRY
    g = GEOMETRY[geo_name]
    return g, dict(max_imgs=T, N=g['N'], split_method='event_count',
                   convert_method='event_histogram', grayscale=grayscale,
                   count_non_zero=g['count_non_zero'], background_mask=g['background_mask'])


def test_config1_full_size_properties(hip):
    """configs[1] at BASELINE size (256 samples x 10 views = 2560 frames, ViT-L/14 full depth) through
    size-independent properties: a sample's outputs do not depend on what else is in the batch
    (the same rows come back BIT-identical from a 3-sample batch, which is the size the oracle
    checks), duplicated samples give identical rows, ragged view counts give the right masks, and
    the probabilities are distributions."""
    import torch
    from eventclip_amd import clip as eclip
    from eventclip_amd.clip_cls import ZSCLIPClassifier
    from eventclip_amd.event2img import Event2ImagePipeline
    from eventclip_amd.synthetic import make_events
    g, qa = quantize_args('n_caltech', 10, grayscale=False)
    cfg = eclip.arch_config('ViT-L/14', text_layers=2)
    sd = eclip.random_state_dict(cfg, seed=5)
    m = eclip.CLIP(cfg, sd, chunk=2560).cuda().eval()
    K = 101
    model = ZSCLIPClassifier(clip_dict=dict(clip_model=m, prompt='a point cloud image of a {}',
                                            class_names=[str(i) for i in range(K)], agg_func='mean',
                                            class_tokens=eclip.synthetic_tokens(K, seed=2))).cuda().eval()
    pipe = Event2ImagePipeline(g['resolution'], g['max_n'], qa, n_px=224, patch=14, kpad=m.kpad)
    N = g['N']
    uniq = [make_events(10 * N, g['resolution'], seed=100 + i) for i in range(6)]
    ragged = [make_events(n, g['resolution'], seed=200 + i) for i, n in enumerate((N // 3, 3 * N + N // 2 + 1, 7 * N))]
    batch = [uniq[i % 6] for i in range(253)] + ragged            # 256 samples, 2530 + 1 + 4 + 7 frames
    out = model(pipe(batch))
    vm = out['valid_masks']
    assert vm.shape == (256, 10) and int(vm.sum()) == 2530 + 1 + 4 + 7
    assert vm[253].tolist() == [True] + [False] * 9 and int(vm[254].sum()) == 4 and int(vm[255].sum()) == 7
    # duplicates of the same sample agree bit for bit wherever they sit in the batch
    for i in range(6):
        rows = out['logits'][i:253:6]
        assert torch.equal(rows, rows[:1].expand_as(rows))
    # ... and equal what a small batch of only those samples produces
    small = model(pipe([uniq[0], ragged[1], uniq[5]]))
    for k in ('logits', 'probs', 'full_logits'):
        assert torch.equal(small[k][0], out[k][0]) and torch.equal(small[k][1], out[k][254])
        assert torch.equal(small[k][2], out[k][5])
    p = out['probs']
    assert torch.isfinite(out['logits']).all() and float((p.sum(-1) - 1).abs().max()) < 1e-5
    assert float(out['full_logits'][~vm].abs().max()) == 0.          # clip_cls.py:151-152


@WEIGHTS
def test_config1_ncaltech_rgb_vitl14_full_depth(hip, weights):
    """configs[1] (the bench workload: N-Caltech101 zero-shot, ViT-L/14, RGB polarity, 10 views) at full
    depth against the fp32 oracle chain, ragged view counts included."""
    import torch
    inp, out, want, pipe = run_config(1, weights)
    assert want['valid_masks'].sum(1).tolist() == [10, 2, 6]
    assert torch.equal(out['logits'].argmax(-1).cpu(), want['logits'].argmax(-1))


@pytest.mark.parametrize('mode', ['plain_chain', 'precise', 'first2', 'first4', 'f16_weights'])
def test_config1_signal_weights_other_tower_modes(hip, mode):
    """configs[1] on the input-dependent weights through the two other forms of the image tower, so that what the
    round-3 defaults (LayerNorm folded into the GEMMs, pre-scaled q) contribute to the error stays visible:
      plain_chain  CLIP(ln_folded=False, q_scaled=False): fp32 residual stream, LayerNorm launches, in-kernel q scale;
                   same bound as the default path;
      precise      ec_vit_weights.precise (hi + lo operands in every GEMM, 3 x the MFMA work): north_star's 1e-3
                   holds on these weights too -- the measured price of that tolerance is bench.py --precise;
      first2 / first4   ec_vit_weights.precise_blocks: only the first 2 / 4 blocks as split-operand blocks (an early block's
                   rounding error is carried through every later block); held to the default path's bound, the lines
                   show what each count buys (bench.py --precise-blocks N prices it);
      f16_weights  the default tower on the same weights ROUNDED TO 16 BIT FIRST (oracle included; config_cases
                   weights='signal16'): what a released checkpoint is -- clip.load() on a GPU returns fp16 parameters
                   (reference test.py:25-26) -- so the rounding of the fp32 random weights, which the other cases count
                   as the HIP path's error, is not there: 1.40e-3 instead of 1.92e-3 (same bound as the default path)."""
    import torch
    key = 'n_caltech/ViT-L/14'
    kw = dict(ln_folded=False, q_scaled=False) if mode == 'plain_chain' else dict(image_precise=True) if mode == 'precise' \
        else {} if mode == 'f16_weights' else dict(image_precise_blocks=int(mode[5:]))
    inp = build_inputs(1, 'signal16' if mode == 'f16_weights' else 'signal', 0)
    out, _, _ = hip_case(inp, **kw)
    want, feats, _, _ = oracle_for(inp, need_emu=False)
    e = logit_errors({k: v.cpu() for k, v in out.items()}, want)
    line = (f'[configs[1] {mode}] full_logits error vs the fp32 oracle, max-normalised / centred: '
            f'{e["full_logits"][0]:.2e} / {e["full_logits"][1]:.2e}; aggregated logits {e["logits"][0]:.2e} / {e["logits"][1]:.2e}')
    print('\n' + line)
    record_parity(line)
    tol = LOGIT_TOL if mode == 'precise' else SIGNAL_TOL[key]
    assert signal_share(feats) >= SIGNAL_GAINS[key][2]
    assert e['full_logits'][0] < tol and e['logits'][0] < tol, e
    assert torch.equal(out['logits'].argmax(-1).cpu(), want['logits'].argmax(-1))


@WEIGHTS
def test_config0_ncaltech_gray_vitb32_batch1(hip, weights):
    """configs[0]: N-Caltech101 zero-shot, ViT-B/32 (full depth), gray event2img, batch = 1.  The oracle chain runs
    LIVE on this host (12 blocks, 5 frames: seconds) and must reproduce the shipped outputs of the build container:
    the shipped logits of the other configs are then this oracle's, not a stale file's."""
    import torch
    inp, out, want, pipe = run_config(0, weights, live=True)
    assert pipe.max_imgs == 10
    assert int(want['valid_masks'].sum()) == 5                     # 4 chunks + overlap chunk = 5 views
    assert torch.equal(out['logits'].argmax(-1).cpu(), want['logits'].argmax(-1))
    gold = load_golden(0, weights, 0)
    assert gold is not None, 'tests/golden/configs_oracle_0_*.npz missing: python tools/make_golden_configs.py'
    np.testing.assert_allclose(fingerprint(inp), gold['fingerprint'], rtol=1e-9)
    mag = float(want['full_logits'].abs().max())
    assert float(np.abs(want['full_logits'].numpy() - gold['full_logits']).max()) < 2e-5 * mag   # fp32 GEMM order


@WEIGHTS
def test_config2_ncars_fewshot_adapter_vitl14(hip, weights):
    """configs[2]: N-Cars few-shot with the text-trans adapter, ViT-L/14 (all 24 blocks), one
    short view per sample (12 500 < N = 30 000 events), count_non_zero, no background mask."""
    inp, out, want, pipe = run_config(2, weights)
    assert pipe.max_imgs == 1                                      # round(12500 / 30000) = 0 -> 1
    assert want['valid_masks'].shape == (6, 1)


@WEIGHTS
def test_config3_nimagenet_vitl14_336_k1000(hip, weights):
    """configs[3]: N-ImageNet zero-shot, ViT-L/14@336px (all 24 blocks, S = 577), 1000 classes, two views
    of 70 000 events on the 480 x 640 sensor (multi-band, uncached events path)."""
    inp, out, want, pipe = run_config(3, weights)
    assert pipe.max_imgs == 2
    assert want['valid_masks'].tolist() == [[True, True], [True, False]]
    top5 = out['logits'].topk(5, dim=-1).indices.cpu()              # test.py:76-81 top-5 path
    assert all(int(want['logits'][b].argmax()) in top5[b].tolist() for b in range(2))


@WEIGHTS
def test_config4_nimagenet_fewshot_t5_k1000(hip, weights):
    """configs[4]: N-ImageNet few-shot adapter, ViT-L/14 (all 24 blocks), T = 5 views, 1000
    classes, residual 0.95; ragged view counts.  (The reference derives max_imgs from max_n:
    round(135000 / 70000) = 2; T = 5 needs max_n = 350 k.)"""
    inp, out, want, pipe = run_config(4, weights)
    assert pipe.max_imgs == 5
    assert want['valid_masks'].sum(1).tolist() == [5, 2, 1]


def test_shipped_oracle_is_reproduced_live_on_a_full_depth_vitl14(hip):
    """One full-depth ViT-L/14 oracle chain per suite run stays LIVE on the GPU box's host (configs[2], 6 frames, the
    few-shot tail included): it must reproduce tests/golden/configs_oracle_2_signal.npz draw 0."""
    inp = build_inputs(2, 'signal', 0)
    gold = load_golden(2, 'signal', 0)
    assert gold is not None
    np.testing.assert_allclose(fingerprint(inp), gold['fingerprint'], rtol=1e-9)
    want, feats = oracle_case(inp)
    mag = float(np.abs(gold['full_logits']).max())
    assert float(np.abs(want['full_logits'].numpy() - gold['full_logits']).max()) < 2e-5 * mag
    assert float(np.abs(feats.numpy() - gold['feats']).max()) < 2e-5 * float(np.abs(gold['feats']).max())


# ------------------------------------------------------------------------------------------------
# configs[2..4] at their per-GPU BASELINE sizes and full depth, through size-independent properties
# (the oracle checks the same configs above at a handful of samples): a sample's rows do not depend on
# the rest of the batch -- bit for bit the rows of a small batch --, duplicates agree, masks follow the
# event counts, padded views are zero, the probabilities are distributions.
# ------------------------------------------------------------------------------------------------
def _fs_model(m, K, residual, seed):
    import torch
    from eventclip_amd import clip as eclip
    from eventclip_amd.clip_cls import FSCLIPClassifier
    torch.manual_seed(seed)
    model = FSCLIPClassifier(
        adapter_dict=dict(adapter_type='text-trans', in_dim=768, d_model=256, num_heads=4,
                          ffn_dim=1024, norm_first=True, num_layers=2, residual=residual),
        clip_dict=dict(clip_model=m, prompt='a point cloud image of a {}',
                       class_names=[str(i) for i in range(K)], agg_func='mean',
                       class_tokens=eclip.synthetic_tokens(K, seed=seed)),
        loss_dict=dict(use_logits_loss=True, use_probs_loss=False))
    with torch.no_grad():
        for p in model.adapter.parameters():
            p.add_(torch.randn_like(p) * 0.02)
    return model.cuda().eval()


def _batch_properties(model, pipe, batch, uniq_n, small_ids, frames_expected):
    import torch
    out = model(pipe(batch))
    vm = out['valid_masks']
    B = len(batch)
    assert vm.shape[0] == B and int(vm.sum()) == frames_expected
    n_dup = B - 3
    for i in range(uniq_n):                      # duplicates agree bit for bit wherever they sit
        rows = out['logits'][i:n_dup:uniq_n]
        assert torch.equal(rows, rows[:1].expand_as(rows))
    small = model(pipe([batch[i] for i in small_ids]))
    for k in ('logits', 'probs', 'full_logits'):
        for j, i in enumerate(small_ids):
            assert torch.equal(small[k][j], out[k][i]), (k, i)
    p = out['probs']
    assert torch.isfinite(out['logits']).all() and float((p.sum(-1) - 1).abs().max()) < 1e-5
    if (~vm).any():
        assert float(out['full_logits'][~vm].abs().max()) == 0.      # clip_cls.py:151-152 / :329
    return out


def test_config2_full_size_properties(hip):
    """configs[2]: N-Cars few-shot adapter, ViT-L/14 full depth, 512 samples x 1 view."""
    from eventclip_amd import clip as eclip
    from eventclip_amd.event2img import Event2ImagePipeline
    from eventclip_amd.synthetic import make_events
    g, qa = quantize_args('n_cars', 2)
    cfg = eclip.arch_config('ViT-L/14', text_layers=1)
    m = eclip.CLIP(cfg, eclip.random_state_dict(cfg, seed=42), chunk=2560).cuda().eval()
    model = _fs_model(m, 2, 0.8, seed=2)
    pipe = Event2ImagePipeline(g['resolution'], g['max_n'], qa, n_px=224, patch=14, kpad=m.kpad)
    uniq = [make_events(12500, g['resolution'], seed=300 + i) for i in range(7)]
    ragged = [make_events(n, g['resolution'], seed=400 + i) for i, n in enumerate((40, 12500, 29999))]
    batch = [uniq[i % 7] for i in range(509)] + ragged             # every sample < N events: one view
    out = _batch_properties(model, pipe, batch, 7, [0, 510, 6, 509], 512)
    assert out['logits'].shape == (512, 2) and out['valid_masks'].shape == (512, 1)


def test_config3_full_size_properties(hip):
    """configs[3]: N-ImageNet zero-shot, ViT-L/14@336px full depth (S = 577), K = 1000, the per-GPU
    shard of 256 samples x 2 views; the top-5 path of test.py:76-81."""
    import torch
    from eventclip_amd import clip as eclip
    from eventclip_amd.clip_cls import ZSCLIPClassifier
    from eventclip_amd.event2img import Event2ImagePipeline
    from eventclip_amd.synthetic import make_events
    g, qa = quantize_args('n_imagenet', 2)
    cfg = eclip.arch_config('ViT-L/14@336px', text_layers=1)
    m = eclip.CLIP(cfg, eclip.random_state_dict(cfg, seed=43), chunk=2560).cuda().eval()
    K = 1000
    model = ZSCLIPClassifier(clip_dict=dict(clip_model=m, prompt='a point cloud image of a {}',
                                            class_names=[str(i) for i in range(K)], agg_func='mean',
                                            class_tokens=eclip.synthetic_tokens(K, seed=3))).cuda().eval()
    pipe = Event2ImagePipeline(g['resolution'], g['max_n'], qa, n_px=336, patch=14, kpad=m.kpad)
    N = g['N']
    uniq = [make_events(2 * N, g['resolution'], seed=500 + i) for i in range(5)]
    ragged = [make_events(n, g['resolution'], seed=600 + i) for i, n in enumerate((N // 2, N + N // 2 + 1, N))]
    batch = [uniq[i % 5] for i in range(253)] + ragged             # 506 + 1 + 2 + 1 frames
    out = _batch_properties(model, pipe, batch, 5, [1, 253, 254, 4], 510)
    assert out['logits'].shape == (256, K)
    top5 = out['logits'].topk(5, dim=-1).indices
    assert bool((top5[:, 0] == out['logits'].argmax(-1)).all())
    assert out['valid_masks'][253].tolist() == [True, False] and out['valid_masks'][254].tolist() == [True, True]


def test_config4_full_size_properties(hip):
    """configs[4]: N-ImageNet few-shot adapter, ViT-L/14 full depth, K = 1000, the per-GPU shard of
    512 samples x 5 views (2560 frames)."""
    from eventclip_amd import clip as eclip
    from eventclip_amd.event2img import Event2ImagePipeline
    from eventclip_amd.synthetic import make_events
    g, qa = quantize_args('n_imagenet', 5)
    cfg = eclip.arch_config('ViT-L/14', text_layers=1)
    m = eclip.CLIP(cfg, eclip.random_state_dict(cfg, seed=44), chunk=2560).cuda().eval()
    model = _fs_model(m, 1000, 0.95, seed=4)
    pipe = Event2ImagePipeline(g['resolution'], 350000, qa, n_px=224, patch=14, kpad=m.kpad)
    N = g['N']
    uniq = [make_events(5 * N, g['resolution'], seed=700 + i) for i in range(4)]
    ragged = [make_events(n, g['resolution'], seed=800 + i) for i, n in enumerate((N - 1, 2 * N, 3 * N + N // 2 + 1))]
    batch = [uniq[i % 4] for i in range(509)] + ragged             # 2545 + 1 + 2 + 4 frames
    out = _batch_properties(model, pipe, batch, 4, [3, 509, 511, 0], 2552)
    assert out['logits'].shape == (512, 1000) and out['valid_masks'].sum(1)[509:].tolist() == [1, 2, 4]


# draws of tests/config_cases.py the tolerance mode is held to 1e-3 on, per config: the historical pair (0), the WORST of draws
# 1 .. 7 (the draws the counts were picked on) and the worst of the HELD-OUT draws 8 .. 15, at the shipped settings
# (profiles/r6_parity_seeds.txt / r6_parity_seeds_held_out.txt: worst 2.4e-4 / 4.8e-4 / 8.0e-4 / 6.1e-4 and 2.9e-4 / 4.8e-4 / 7.2e-4 /
# 5.5e-4 / 7.2e-4).  configs[2] draw 5 (fp32 weights) is the one pair of 160 outside, held by
# test_tolerance_mode_configs2_small_logits_draw; the worst of configs[2]'s other selection draws is draw 3
TOLERANCE_DRAWS = {0: (0, 3, 14), 1: (0, 4, 12), 2: (0, 3, 12), 3: (0, 6, 9), 4: (0, 2, 9)}


@pytest.mark.parametrize('config,draw', [(c, d) for c in range(5) for d in TOLERANCE_DRAWS[c]])
def test_tolerance_mode_meets_1e3_over_draws(hip, config, draw, monkeypatch):
    """north_star's 1e-3 on input-dependent weights with the tolerance mode (ec_vit_weights.precise_blocks /
    precise_attn_blocks at eventclip_amd.clip.TOLERANCE_MODE's counts: the first blocks of the image tower as
    split-operand blocks, the first few of them with fp32-class attention, the rest as the folded 16-bit chain), on
    THREE (weight seed, event seed) draws per config -- one of them held out when the counts were picked -- against the oracle
    logits shipped in tests/golden/configs_oracle_* (profiles/r6_parity_seeds.txt / _held_out.txt / _16bit_weights.txt have every
    draw, default path and tolerance mode: median, worst case, fraction inside 1e-3)."""
    import sys
    from eventclip_amd import clip as eclip
    mod = sys.modules[__name__]
    monkeypatch.setattr(mod, 'LINE_TAG', ', tolerance mode')
    run_config(config, 'signal', draw, tol=LOGIT_TOL, **eclip.tolerance_mode_kwargs(CASES[config]['arch']))


# ... and on a checkpoint stored in 16 bit (the kind the mode is priced on: no weight lo products): per config the worst of the
# sixteen draws on weights rounded to 16 bit first (profiles/r6_parity_seeds_16bit_weights.txt / _16bit_weights_held_out.txt:
# 2.1e-4 / 5.9e-4 / 7.9e-4 / 6.2e-4 / 7.4e-4)
TOLERANCE_DRAWS_16BIT = {0: 12, 1: 12, 2: 5, 3: 9, 4: 12}


@pytest.mark.parametrize('config', range(5))
def test_tolerance_mode_meets_1e3_on_16bit_checkpoints(hip, config, monkeypatch):
    """The same bound on weights rounded to 16 bit first ('signal16': what a released CLIP checkpoint is; CLIP._pack finds
    the lo parts zero and sets ec_vit_weights.weights_exact16), on the worst measured draw of each config."""
    import sys
    from eventclip_amd import clip as eclip
    mod = sys.modules[__name__]
    monkeypatch.setattr(mod, 'LINE_TAG', ', tolerance mode, 16-bit weights')
    run_config(config, 'signal16', TOLERANCE_DRAWS_16BIT[config], tol=LOGIT_TOL, **eclip.tolerance_mode_kwargs(CASES[config]['arch']))


def test_tolerance_mode_configs2_small_logits_draw(hip):
    """The one (config, draw) of the 160 in profiles/r6_parity_seeds*.txt on which the tolerance mode is NOT inside 1e-3:
    configs[2] draw 5 (two classes, few-shot head: logits = 100 cos).  Its largest |logit| is 1.53 -- every cosine of the
    batch below 0.016, five times smaller than the other draws' -- so the same absolute error (2e-3 logit units = 2e-5 in
    the cosine) reads as 1.3e-3 of max |logit| (7.9e-4, inside, on the same draw with the weights rounded to 16 bit).  Held here at what it measures, so that the number in the docs stays true:
    1.3 - 1.9e-3 in the mode by setting (default path: 2.5e-3): below 2.5e-3, absolute error below 4e-3 logit units, classes ranked as the oracle
    ranks them wherever the oracle separates them by more than the error."""
    import torch
    from eventclip_amd import clip as eclip
    inp = build_inputs(2, 'signal', 5)
    out, _, _ = hip_case(inp, **eclip.tolerance_mode_kwargs(CASES[2]['arch']))
    want, feats, _, _ = oracle_for(inp, need_emu=False)
    e = logit_errors({k: v.cpu() for k, v in out.items()}, want)
    mag = float(want['full_logits'].abs().max())
    line = f'[configs[2] draw 5, tolerance mode] max |logit| {mag:.2f}; full_logits error {e["full_logits"][0]:.2e} of it = {e["full_logits"][0] * mag:.2e} logit units'
    print('\n' + line)
    record_parity(line)
    assert mag < 2.0                                               # the denominator that makes this draw what it is
    assert e['full_logits'][0] < 2.5e-3 and e['full_logits'][0] * mag < 4e-3
    assert ranks_agree(out['logits'].cpu(), want['logits'], 1, 2 * e['logits'][0] * mag)
