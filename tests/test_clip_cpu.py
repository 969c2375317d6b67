"""Host-side behaviour of the CLIP wrapper and of the new C entry points that do not need a GPU."""
import warnings

import pytest


def test_precise_blocks_env_is_reported_when_it_cannot_apply(monkeypatch):
    """EVENTCLIP_PRECISE_BLOCKS addresses every model of the process; where the mode cannot apply (too many blocks, bf16,
    the plain chain, low latency) the model says so instead of silently running the 16-bit path (advisor, round 4), and
    the explicit argument keeps its own error path in _pack."""
    from eventclip_amd import clip as eclip
    cfg = eclip.arch_config('ViT-B/32', layers=4, text_layers=1, vocab_size=64)
    sd = eclip.random_state_dict(cfg, seed=0)
    monkeypatch.setenv('EVENTCLIP_PRECISE_BLOCKS', '8')            # 8 >= 4 layers
    with pytest.warns(UserWarning, match='EVENTCLIP_PRECISE_BLOCKS=8 ignored'):
        m = eclip.CLIP(cfg, sd)
    assert m.image_precise_blocks == 0 and m.image_precise_attn_blocks == 0
    monkeypatch.setenv('EVENTCLIP_PRECISE_BLOCKS', '2')
    with pytest.warns(UserWarning, match='ignored'):
        assert eclip.CLIP(cfg, sd, dtype='bfloat16').image_precise_blocks == 0
    with pytest.warns(UserWarning, match='ignored'):
        assert eclip.CLIP(cfg, sd, ln_folded=False).image_precise_blocks == 0
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        m = eclip.CLIP(cfg, sd)                                     # applies: no warning
    assert m.image_precise_blocks == 2 and m.image_precise_attn_blocks == 2          # min(default 5, 2)
    big = eclip.arch_config('ViT-L/14@336px', layers=10, text_layers=1, vocab_size=64)
    monkeypatch.setenv('EVENTCLIP_PRECISE_BLOCKS', '8')
    assert eclip.CLIP(big, eclip.random_state_dict(big, seed=0)).image_precise_attn_blocks == 7       # 577 tokens: seven
    monkeypatch.setenv('EVENTCLIP_PRECISE_BLOCKS', '2')
    monkeypatch.setenv('EVENTCLIP_PRECISE_ATTN_BLOCKS', '1')
    assert eclip.CLIP(cfg, sd).image_precise_attn_blocks == 1
    monkeypatch.delenv('EVENTCLIP_PRECISE_BLOCKS')
    assert eclip.CLIP(cfg, sd).image_precise_blocks == 0
    assert eclip.CLIP(cfg, sd, image_precise_blocks=3, image_precise_attn_blocks=9).image_precise_attn_blocks == 3


def test_classify_workspace_size_is_a_host_function():
    """ec_classify_v2_workspace_bytes / ec_classify_text_bytes carve the hi / lo operand planes and the raw product; no
    device call, so they answer on a CPU-only box (the call needs the rows' planes, the text planes are prepared once)."""
    from eventclip_amd import _lib
    lib = _lib.lib()
    n, C, K = 2560, 768, 101
    b = int(lib.ec_classify_v2_workspace_bytes(n, C, K))
    assert b >= 2 * n * C * 2 + n * 112 * 4 + n * 4
    assert b % 256 == 0 and int(lib.ec_classify_v2_workspace_bytes(0, C, K)) == 0
    t = int(lib.ec_classify_text_bytes(C, K))
    assert t >= 2 * 112 * C * 2 + 112 * 4 and t % 256 == 0 and int(lib.ec_classify_text_bytes(0, K)) == 0


def test_tolerance_mode_settings(monkeypatch):
    """eventclip_amd.clip.TOLERANCE_MODE is the one place the tolerance mode's counts live (bench.py prices them, the config tests
    hold them to 1e-3, profiles/r6_parity_seeds.json records them): by sequence length, clamped to the depth of the tower."""
    from eventclip_amd import clip as eclip
    short, long_ = eclip.TOLERANCE_MODE
    assert eclip.tolerance_mode_kwargs('ViT-L/14') == dict(image_precise_blocks=short[0], image_precise_attn_blocks=short[1])
    assert eclip.tolerance_mode_kwargs('ViT-L/14@336px') == dict(image_precise_blocks=long_[0], image_precise_attn_blocks=long_[1])
    kw = eclip.tolerance_mode_kwargs('ViT-B/32')                                  # 12 layers: at most 11 split-operand blocks
    assert kw['image_precise_blocks'] == min(short[0], 11) and kw['image_precise_attn_blocks'] <= kw['image_precise_blocks']
    cfg = eclip.arch_config('ViT-B/32', layers=4, text_layers=1, vocab_size=64)
    sd = eclip.random_state_dict(cfg, seed=0)
    m = eclip.CLIP(cfg, sd, **eclip.tolerance_mode_kwargs(cfg))
    assert m.image_precise_blocks == 3 and m.image_lo_fp8 == eclip.DEFAULT_LO_FP8
    assert eclip.CLIP(cfg, sd).image_lo_fp8 is False                              # no split-operand blocks: nothing to run on e4m3
    monkeypatch.setenv('EVENTCLIP_LO_FP8', '0')
    assert eclip.CLIP(cfg, sd, image_precise_blocks=2).image_lo_fp8 is False
    monkeypatch.setenv('EVENTCLIP_TOLERANCE_MODE', '2:1')
    assert eclip.tolerance_mode_kwargs(cfg) == dict(image_precise_blocks=2, image_precise_attn_blocks=1)
