#!/usr/bin/env python3
"""GPU idle time inside hipGraph replays of the train step: from a rocprofv3 kernel trace (csv) of `bench.py --steps K`, the union of the
kernels' busy intervals over the last replays against their span.  usage: python3 scripts/graph_gaps.py <kernel_trace.csv> [replays]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# the replays are the last k occurrences of the step's one adamw_flat_kernel launch
ends = [e for s, e, n in iv if "adamw_flat_kernel" in n]
if len(ends) < k + 1:
    sys.exit("not enough steps in the trace")
for i in range(len(ends) - k, len(ends)):
    lo, hi = ends[i - 1], ends[i]
    step = [(s, e, n) for s, e, n in iv if s >= lo and e <= hi]
    busy, cur_s, cur_e = 0, None, None
    for s, e, n in step:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    span = step[-1][1] - step[0][0]
    gaps = sorted(((b[0] - a[1]) for a, b in zip(step, step[1:]) if b[0] > a[1]), reverse=True)
    print(f"replay {i}: {len(step)} kernels, span {span / 1e6:.3f} ms, busy (union) {busy / 1e6:.3f} ms, idle {(span - busy) / 1e6:.3f} ms, "
          f"sum of durations {sum(e - s for s, e, n in step) / 1e6:.3f} ms; gaps > 5 us: {sum(1 for g in gaps if g > 5000)}, largest {gaps[:5]}")
