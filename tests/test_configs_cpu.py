"""The shipped oracle outputs of the BASELINE config cases (tests/golden/configs_oracle_*.npz, written by
tools/make_golden_configs.py in the build container) against the oracle chain run HERE: the files are this oracle's
outputs on these inputs, every draw is there, and the inputs regenerate to the recorded fingerprints."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import config_cases as cc   # noqa: E402


def test_every_config_ships_its_draws():
    for c in range(5):
        for weights, draws in (('signal', range(cc.N_DRAWS)), ('init', [0])):
            for d in draws:
                g = cc.load_golden(c, weights, d)
                assert g is not None, (c, weights, d)
                B, T = g['valid_masks'].shape
                K = cc.CASES[c]['K']
                assert g['full_logits'].shape == (B, T, K) and g['logits'].shape == (B, K)
                assert g['emu_full_logits'].shape == (B, T, K) and g['feats'].shape[0] == int(g['valid_masks'].sum())
                assert np.isfinite(g['full_logits']).all() and g['fingerprint'].shape == (6,)
    assert cc.load_golden(1, 'signal16', 0) is not None
    all_draws = list(range(cc.N_DRAWS)) + list(cc.HELD_OUT_DRAWS)
    seeds = {cc.draw_seeds(c, d) for c in range(5) for d in all_draws}
    assert len(seeds) == 5 * len(all_draws)                   # no (weights, events) pair is used twice
    assert len({s[0] for s in seeds}) == len(seeds) and len({s[1] for s in seeds}) == len(seeds)
    for c in range(5):                                        # the held-out draws (never used to pick a setting) ship too
        for d in cc.HELD_OUT_DRAWS:
            assert cc.load_golden(c, 'signal', d) is not None, (c, d)
        for d in all_draws:                                   # ... and every draw on weights rounded to 16 bit
            assert cc.load_golden(c, 'signal16', d) is not None, (c, d)


def test_configs0_oracle_reproduces_the_shipped_outputs():
    """ViT-B/32, 5 frames: the one config whose oracle chain is cheap enough for the CPU suite, on two draws and both
    weight kinds; the fp32 chain within GEMM summation order, the fp16-reference emulation likewise."""
    for weights, d in (('signal', 0), ('signal', 5), ('init', 0)):
        inp = cc.build_inputs(0, weights, d)
        g = cc.load_golden(0, weights, d)
        np.testing.assert_allclose(cc.fingerprint(inp), g['fingerprint'], rtol=1e-9)
        want, feats = cc.oracle_case(inp)
        mag = float(np.abs(g['full_logits']).max())
        assert np.array_equal(want['valid_masks'].numpy(), g['valid_masks'])
        assert float(np.abs(want['full_logits'].numpy() - g['full_logits']).max()) < 2e-5 * mag
        assert float(np.abs(want['logits'].numpy() - g['logits']).max()) < 2e-5 * mag
        assert float(np.abs(feats.numpy() - g['feats']).max()) < 2e-5 * float(np.abs(g['feats']).max())
    emu, _ = cc.oracle_case(cc.build_inputs(0, 'signal', 0), emulate='fp16_reference')
    g = cc.load_golden(0, 'signal', 0)
    # (rounded arithmetic: a summation-order difference can move a value by an fp16 ulp of an intermediate)
    assert float(np.abs(emu['full_logits'].float().numpy() - g['emu_full_logits']).max()) < 2e-3 * float(np.abs(g['full_logits']).max())


def test_adapter_state_is_reproducible():
    a = cc.make_adapter_state(7, 0.8)
    b = cc.make_adapter_state(7, 0.8)
    assert sorted(a) == sorted(b) and all(bool((a[k] == b[k]).all()) for k in a)
    c = cc.make_adapter_state(8, 0.8)
    assert any(not bool((a[k] == c[k]).all()) for k in a)
