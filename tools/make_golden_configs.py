"""fp32 oracle logits of the BASELINE config cases, computed ONCE in the build container and shipped as
tests/golden/configs_oracle_{config}_{weights}.npz (the 1e-3 claim on a distribution of
(weights, events) draws, and the full-depth fp32 CPU towers out of the GPU box's lease).

    python tools/make_golden_configs.py                      # every config, 'signal' x 8 draws + 'init' x 1 draw
    python tools/make_golden_configs.py --configs 1 --draws 0,1 --weights signal

Runs the builder's own oracle chain (oracle/events.py -> oracle/preprocess.py -> oracle/clip_ref.py -> oracle/classify.py
/ oracle/adapter.py; tests/config_cases.py:oracle_case) on the cases tests/config_cases.py defines, in fp32 and in the
fp16-reference emulation (the yardstick).  Nothing of /root/reference is read: the oracle modules are restatements whose
own pins are the other make_golden_* scripts; clip_ref.py stays PARITY UNPINNED against openai/CLIP (its header).
Per draw the file holds: full_logits, logits, valid_masks, feats (image features), emu_full_logits, emu_logits and the
fingerprint of the regenerated inputs.  Existing draws in a file are kept (the run is resumable).
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    import config_cases as cc
    ap = argparse.ArgumentParser()
    ap.add_argument('--configs', default='0,1,2,3,4')
    ap.add_argument('--weights', default='signal,init')
    ap.add_argument('--draws', default=None, help="default: 0..7 for 'signal', 0 for 'init'")
    ap.add_argument('--force', action='store_true', help='recompute draws that are already in the file')
    a = ap.parse_args()
    for c in [int(x) for x in a.configs.split(',')]:
        for weights in a.weights.split(','):
            draws = [int(x) for x in a.draws.split(',')] if a.draws else (list(range(cc.N_DRAWS)) if weights.startswith('signal') else [0])
            path = cc.golden_path(c, weights)
            z = dict(np.load(path, allow_pickle=False)) if os.path.exists(path) else {}
            for d in draws:
                if f'd{d}_full_logits' in z and not a.force:
                    continue
                t0 = time.time()
                inp = cc.build_inputs(c, weights, d)
                want, feats = cc.oracle_case(inp)
                emu, _ = cc.oracle_case(inp, emulate='fp16_reference')
                z[f'd{d}_full_logits'] = want['full_logits'].numpy().astype(np.float32)
                z[f'd{d}_logits'] = want['logits'].numpy().astype(np.float32)
                z[f'd{d}_valid_masks'] = want['valid_masks'].numpy()
                z[f'd{d}_feats'] = feats.numpy().astype(np.float32)
                z[f'd{d}_emu_full_logits'] = emu['full_logits'].float().numpy()
                z[f'd{d}_emu_logits'] = emu['logits'].float().numpy()
                z[f'd{d}_fingerprint'] = cc.fingerprint(inp)
                z[f'd{d}_seeds'] = np.asarray([inp['wseed'], inp['eseed']], np.int64)
                np.savez_compressed(path, **z)
                share = float((feats - feats.mean(0)).norm() / feats.norm())
                print(f'configs[{c}] {weights} draw {d} (seeds {inp["wseed"]}, {inp["eseed"]}): {feats.shape[0]} frames, '
                      f'share {share:.2f}, max |logit| {float(want["full_logits"].abs().max()):.2f}, {time.time() - t0:.0f} s',
                      flush=True)


if __name__ == '__main__':
    main()
