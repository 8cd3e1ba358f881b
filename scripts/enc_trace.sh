#!/bin/bash
# Per-dispatch kernel trace of the encoders' forward + backward (scripts/enc_micro.py, mode 1) -> gpurun_out/enc_trace.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MIOPEN_FIND_MODE=2 ENC_MODES=1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_enc -- python3 scripts/enc_micro.py > gpurun_out/enc_trace.log 2>&1
f=$(find gpurun_out/prof_enc -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:90]
    key = (name, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"])
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg.setdefault(key, [0, 0]); a[0] += 1; a[1] += d
tot = sum(a[1] for a in agg.values())
out = open("gpurun_out/enc_trace.csv", "w")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    out.write(f"{a[1] / tot * 100:5.1f}%  n={a[0]:4d}  avg {a[1] / a[0] / 1e3:8.1f} us  grid {k[1]}x{k[2]}x{k[3]} wg {k[4]}  {k[0]}\n")
PY
find gpurun_out/prof_enc -name "*.csv" -delete; find gpurun_out/prof_enc -name "*.db" -delete
head -60 gpurun_out/enc_trace.csv
