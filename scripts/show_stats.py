#!/usr/bin/env python3
"""Per-step table of a rocprofv3 kernel_stats csv (scripts/prof.sh profiles 3 eager steps): `show_stats.py file.csv [steps] [rows]`."""
import csv
import sys

path = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
nrows = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows = list(csv.DictReader(open(path)))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"kernel time per step: {tot / steps / 1e6:.3f} ms, {sum(int(r['Calls']) for r in rows) // steps} launches")
for r in rows[:nrows]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:78]
    print(f"{n:78s} {int(r['Calls']) // steps:5d} {int(r['TotalDurationNs']) / steps / 1e6:8.3f} ms {float(r['AverageNs']) / 1e3:8.1f} us")
