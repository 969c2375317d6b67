"""HBM traffic of the tolerance mode's own kernels (round 6): fold the rocprofv3 FETCH_SIZE / WRITE_SIZE passes over
`bench.py --tolerance-mode --f16-weights` into one table -- per kernel (template arguments kept where they tell the forms apart)
the fabric-side bytes per full-size launch, and for the two kernels the mode adds (ec_layernorm_hl8, attention_hl_kernel)
the algorithmic bytes next to them.

    python tools/tolerance_traffic.py FETCH_counter_collection.csv WRITE_counter_collection.csv [frames] > profiles/r6_tolerance_traffic.json

Units as tools/traffic_summary.py: both counters in KiB, FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md, HBM / rocprofv3).
"""
import csv
import json
import re
import subprocess
import sys
from collections import defaultdict


def short(name):
    if name.startswith('_Z'):
        try:
            name = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', name], capture_output=True, text=True).stdout.strip() or name
        except OSError:
            pass
    if name.startswith('_Z'):      # cxxfilt gives up on _Float16 parameters: kernel name + integral template arguments by hand
        m = re.search(r'\d+([a-z0-9_]+_kernel)(?:I((?:L[ib]\d+E)+)E)?', name)
        if m:
            args = [(t == 'b' and ('true' if v == '1' else 'false')) or v for t, v in re.findall(r'L([ib])(\d+)E', m.group(2) or '')]
            return m.group(1) + (f"<{', '.join(args)}>" if args else '')
    name = name.replace('(anonymous namespace)::', '')
    if name.startswith('void '):
        name = name[5:]
    return re.sub(r'\(.*$', '', name).strip()


def collect(path, counter):
    rows = defaultdict(list)
    for row in csv.DictReader(open(path)):
        if row['Counter_Name'] == counter:
            rows[short(row['Kernel_Name'])].append((int(row['Grid_Size']), float(row['Counter_Value'])))
    out = {}
    for k, v in rows.items():
        gmax = max(g for g, _ in v)
        vals = [c for g, c in v if g == gmax]
        vals = [c for c in vals if c > 0.25 * max(vals)] if max(vals) > 0 else vals
        out[k] = (sum(vals) / len(vals), len(vals))
    return out


def main():
    fetch, write = collect(sys.argv[1], 'FETCH_SIZE'), collect(sys.argv[2], 'WRITE_SIZE')
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 2560
    S, width, heads, hd = 257, 1024, 16, 64
    alg = {
        # hi + lo planes in (2 + 2 B), 16-bit hi part + e4m3 lo part out (2 + 1 B) per element
        'layernorm_kernel<0, true, true>': frames * S * width * 7,
        # q, k, v as hi + lo parts in, the output as hi + lo parts out: 16 B per (token, head, channel)
        'attention_hl_kernel': frames * heads * S * hd * 16,
    }
    table = {}
    for k in sorted(set(fetch) & set(write)):
        f, nf = fetch[k]
        w, nw = write[k]
        hbm = (2 * f + w) * 1024
        if hbm < 64e6:
            continue
        e = {'FETCH_SIZE_KiB_avg': f, 'WRITE_SIZE_KiB_avg': w, 'full_size_launches': nf, 'hbm_bytes_per_launch': hbm}
        for pat, b in alg.items():
            if k.startswith(pat.split('<')[0]) and (('<' not in pat) or pat in k):
                e['algorithmic_bytes_per_launch'], e['ratio'] = b, hbm / b
        table[k] = e
    json.dump({'note': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `bench.py --tolerance-mode --f16-weights '
                       '--steps 1 --warmup 1`; kernels above 64 MB per launch; fabric-side counters (Infinity-Cache hits included)',
               'frames': frames, 'kernels': table}, sys.stdout, indent=1)


if __name__ == '__main__':
    main()
