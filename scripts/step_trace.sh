#!/bin/bash
# Per-dispatch kernel trace of one eager bench step, aggregated by (kernel, grid): which launches of which shape take the time.
# usage (GPU box, repo root): bash scripts/step_trace.sh <tag> [bench.py args]   -> gpurun_out/step_trace_<tag>.txt
tag=$1
shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MIOPEN_FIND_MODE=2
timeout 900 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_st_$tag -- python3 bench.py --steps 2 --warmup 1 --graph 0 --one-stream --no-cpu-baseline --no-kernel-timing --no-extra "$@" > gpurun_out/step_trace_$tag.log 2>&1
f=$(find gpurun_out/prof_st_$tag -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$tag" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 3:]                      # drop the warm-up step (three equal steps in the trace)
agg = collections.OrderedDict()
gap = 0
prev_end = None
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:100]
    key = (name, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"])
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    a = agg.setdefault(key, [0, 0]); a[0] += 1; a[1] += e - s
    if prev_end is not None and s > prev_end:
        gap += s - prev_end
    prev_end = e if prev_end is None else max(prev_end, e)
tot = sum(a[1] for a in agg.values())
out = open(f"gpurun_out/step_trace_{sys.argv[2]}.txt", "w")
out.write(f"# two eager steps: kernel time {tot / 2e6:.2f} ms per step, idle gaps between kernels {gap / 2e6:.2f} ms per step, {len(rows) // 2} launches per step\n")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    wgs = int(k[1]) * int(k[2]) * int(k[3]) // max(int(k[4]), 1)
    out.write(f"{a[1] / tot * 100:5.1f}%  n/step={a[0] / 2:6.1f}  avg {a[1] / a[0] / 1e3:8.1f} us  per step {a[1] / 2e6:6.2f} ms  wgs {wgs:6d} x {k[4]:>4s}  {k[0]}\n")
PY
find gpurun_out/prof_st_$tag -name "*.csv" -delete; find gpurun_out/prof_st_$tag -name "*.db" -delete
head -70 gpurun_out/step_trace_$tag.txt
