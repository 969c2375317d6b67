"""Logit error of the five BASELINE configs (input-dependent weights) for combinations of the tolerance mode's knobs:
EVENTCLIP_PRECISE_BLOCKS (split-operand blocks) x EVENTCLIP_PRECISE_ATTN_BLOCKS (of which fp32 attention).

    python tools/sweep_tolerance.py 8:4 8:2 12:4 > profiles/r5_tolerance_sweep.txt

Runs tests/test_configs_gpu.py's own config cases (same seeds, same oracle) with the bound lifted, prints one line per
(combination, config): the max-normalised error of full_logits / logits against the fp32 oracle.
"""
import os
import re
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    import test_configs_gpu as tc
    fns = [tc.test_config0_ncaltech_gray_vitb32_batch1, tc.test_config1_ncaltech_rgb_vitl14_full_depth,
           tc.test_config2_ncars_fewshot_adapter_vitl14, tc.test_config3_nimagenet_vitl14_336_k1000,
           tc.test_config4_nimagenet_fewshot_t5_k1000]
    configs = [int(c) for c in os.environ.get('SWEEP_CONFIGS', '0,1,2,3,4').split(',')]
    for k in list(tc.SIGNAL_TOL):
        tc.SIGNAL_TOL[k] = 1.0          # no bound: this tool reports
    # the oracle chain of a config does not depend on the HIP mode: computed once per (config, arithmetic)
    cache, plain = {}, tc.oracle_forward

    def cached(evs, geo, qa, cfg, sd, tokens, T, agg, adapter=None, emulate=None):
        key = (tuple(geo), cfg['image_size'], cfg['width'], cfg['layers'], len(evs), sum(len(e) for e in evs), T, adapter is None, emulate)
        if key not in cache:
            cache[key] = plain(evs, geo, qa, cfg, sd, tokens, T, agg, adapter=adapter, emulate=emulate)
        return cache[key]
    tc.oracle_forward = cached
    for combo in sys.argv[1:]:
        pb, pa = combo.split(':')
        os.environ['EVENTCLIP_PRECISE_BLOCKS'], os.environ['EVENTCLIP_PRECISE_ATTN_BLOCKS'] = pb, pa
        for c in configs:
            with tempfile.NamedTemporaryFile('r', suffix='.txt') as f:
                os.environ['EC_PARITY_TABLE'] = f.name
                tc.LINE_TAG = f', precise_blocks = {pb}, precise_attn_blocks = {pa}'
                try:
                    fns[c](None, 'signal')
                    status = 'ok'
                except AssertionError as e:
                    status = 'assert: ' + str(e)[:80].replace('\n', ' ')
                line = f.read().strip()
            m = re.search(r'HIP ([0-9.e+-]+) / ([0-9.e+-]+),.*aggregated logits: HIP ([0-9.e+-]+)', line)
            errs = f'full_logits {m.group(1)} (centred {m.group(2)}) logits {m.group(3)}' if m else line[:120]
            print(f'precise_blocks={pb} attn={pa} configs[{c}]: {errs}  [{status}]', flush=True)


if __name__ == '__main__':
    main()
