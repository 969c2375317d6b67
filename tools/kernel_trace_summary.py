"""Per kernel class, the rocprofv3 --kernel-trace durations of the launches that belong to the bench
STEP (the full-size ones), next to the all-launch averages of `*_kernel_stats.csv`.

    python tools/kernel_trace_summary.py <kernel_trace.csv> > profiles/rN_x_kernel_trace_step.json

`kernel_stats.csv` averages every dispatch of a symbol in the process: for the GEMMs that mixes the
bench step's launches with the text tower's (run once, 77-token sequences) and the class-token-only
launches of the last block.  bench.py's `roofline.avg_launch_ms` is over the timed steps; to compare
like with like this keeps, per class, the dispatches whose duration is within 4x of the class's longest
(the big launches) and reports both sets.
"""
import csv
import json
import sys
from collections import defaultdict

sys.path.insert(0, __file__.rsplit('/', 1)[0])
from traffic_summary import classify  # noqa: E402


def main():
    rows = defaultdict(list)
    for r in csv.DictReader(open(sys.argv[1])):
        name = classify(r['Kernel_Name'])
        if name.startswith('__amd') or 'Cijk' in name:
            continue
        rows[name].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
    out = {}
    for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        big = [d for d in v if d > max(v) / 4]
        out[k] = {'dispatches': len(v), 'avg_ms_all': sum(v) / len(v), 'total_ms': sum(v),
                  'step_sized_dispatches': len(big), 'avg_ms_step_sized': sum(big) / len(big),
                  'max_ms': max(v)}
    json.dump(out, sys.stdout, indent=1)


if __name__ == '__main__':
    main()
