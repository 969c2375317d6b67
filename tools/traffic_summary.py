"""Fold the rocprofv3 FETCH_SIZE / WRITE_SIZE passes over bench.py into profiles/traffic.json.

    python tools/traffic_summary.py FETCH_counter_collection.csv WRITE_counter_collection.csv bench.json [commit] > profiles/traffic.json

Units per MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters are in KiB; on gfx950
FETCH_SIZE counts the 128-byte requests of 16-B/lane streams at 64 B, so it is doubled.
"""
import csv
import json
import re
import sys
from collections import defaultdict

EPI = {0: 'STORE16', 1: 'GELU16', 2: 'RESID32', 3: 'STORE32', 4: 'GELU16', 5: 'STORE16', 6: 'RESID32', 7: 'STORE16', 8: 'GELU16'}   # 6-8: the folded-LayerNorm forms of the same classes


def classify(name):
    m = re.search(r'(gemm2pp_kernel|gemm2p_kernel|gemm4p_kernel|gemm_b2p?_kernel)<(\d+), (\d+)', name)
    if m:
        return f'gemm_kernel<{EPI[int(m.group(3))]}>'
    m = re.search(r'gemm_kernel<(\d+), \d+, \d+, \d+, \d+, (\d+)', name)
    if m:
        return f'gemm_kernel<{EPI[int(m.group(2))]}>'
    if 'events_pack10_kernel' in name or 'events_band10_kernel' in name:      # same launch site as the 32-bit kernel behind them (one class in bench.py)
        return 'events_to_frames_kernel'
    m = re.search(r'::(\w+_kernel)', name)
    return m.group(1) if m else name.split('(')[0]


def collect(path, counter):
    """Per kernel class: mean counter value over its full-size launches (the text tower's few small
    launches of the same kernels, recognisable by their smaller grids or values, are left out)."""
    rows = defaultdict(list)
    for row in csv.DictReader(open(path)):
        if row['Counter_Name'] != counter:
            continue
        rows[classify(row['Kernel_Name'])].append((int(row['Grid_Size']), float(row['Counter_Value'])))
    tot, cnt = {}, {}
    for k, v in rows.items():
        gmax = max(g for g, _ in v)
        vals = [c for g, c in v if g == gmax]
        vmax = max(vals)
        vals = [c for c in vals if c > 0.25 * vmax]
        tot[k], cnt[k] = sum(vals), len(vals)
    return tot, cnt


def summarise(fetch_csv, write_csv, bench_json):
    """{kernel class: counters, HBM bytes per launch, algorithmic bytes, ratio} for one profiled bench command"""
    bench = json.loads([ln for ln in open(bench_json) if ln.startswith('{')][-1])
    alg = dict(bench['kernel_algorithmic_bytes_per_launch'])
    ft, fc = collect(fetch_csv, 'FETCH_SIZE')
    wt, wc = collect(write_csv, 'WRITE_SIZE')
    kernels = {}
    for k in alg:
        if k not in ft or k not in wt or not alg[k]:
            continue
        f, w = ft[k] / fc[k], wt[k] / wc[k]
        hbm = (2 * f + w) * 1024
        kernels[k] = {'FETCH_SIZE_KiB_avg': f, 'FETCH_SIZE_dispatches': fc[k], 'WRITE_SIZE_KiB_avg': w,
                      'WRITE_SIZE_dispatches': wc[k], 'hbm_bytes_per_launch': hbm,
                      'algorithmic_bytes_per_launch': alg[k], 'ratio': hbm / alg[k]}
    return bench, kernels


def main():
    # [--config N fetch.csv write.csv bench.json] ... after the four positional arguments: the same two passes over
    # `bench.py --config N`, folded in under configs[N] (the kernels the other BASELINE configs run: the row-band
    # events kernel and the 480 x 640 -> 336 preprocess of N-ImageNet, attention at S = 577)
    argv = sys.argv[1:]
    extra = []
    while '--config' in argv:
        i = argv.index('--config')
        extra.append(argv[i + 1:i + 5])
        argv = argv[:i] + argv[i + 5:]
    sys.argv = sys.argv[:1] + argv
    fetch_csv, write_csv, bench_json = sys.argv[1:4]
    bench = json.loads([ln for ln in open(bench_json) if ln.startswith('{')][-1])
    alg = dict(bench['kernel_algorithmic_bytes_per_launch'])   # the events kernel's figure includes its 16 B / event
    ft, fc = collect(fetch_csv, 'FETCH_SIZE')
    wt, wc = collect(write_csv, 'WRITE_SIZE')
    kernels = {}
    for k in alg:
        if k not in ft or k not in wt or not alg[k]:
            continue
        f, w = ft[k] / fc[k], wt[k] / wc[k]
        hbm = (2 * f + w) * 1024
        kernels[k] = {'FETCH_SIZE_KiB_avg': f, 'FETCH_SIZE_dispatches': fc[k], 'WRITE_SIZE_KiB_avg': w,
                      'WRITE_SIZE_dispatches': wc[k], 'hbm_bytes_per_launch': hbm,
                      'algorithmic_bytes_per_launch': alg[k], 'ratio': hbm / alg[k]}
    dom = bench['roofline']['kernel']
    out = {'kernel': dom,
           'commit': sys.argv[4] if len(sys.argv) > 4 else 'unrecorded',   # tree the counters were taken on
           'note': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over '
                   '`bench.py --steps 1 --warmup 1 --no-cpu-baseline`, averaged over the full-size launches '
                   'of each kernel class. Units KiB; '
                   'FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950). Fabric-side counters: '
                   'Infinity-Cache hits are included.'}
    out.update(kernels.get(dom, {}))
    out['all_kernels'] = kernels
    if extra:
        out['configs'] = {}
        for cfg, f_csv, w_csv, b_json in extra:
            b, ks = summarise(f_csv, w_csv, b_json)
            out['configs'][cfg] = {'workload': b['config']['workload'], 'kernels': ks}
    json.dump(out, sys.stdout, indent=1)


if __name__ == '__main__':
    main()
