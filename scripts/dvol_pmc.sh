#!/bin/bash
# Where the gradient-volume kernels spend their cycles (VERDICT r3 next #3a): one PMC pass with the kernel trace over
# scripts/dvol_micro.py -- LDS array cycles and bank-conflict cycles, VALU / LDS / wait cycles per wave, waves -- for the
# row-segment kernel, the round-3 bounding-box kernel and the round-4 separable kernel.
# usage (GPU box, repo root): bash scripts/dvol_pmc.sh <tag>    -> gpurun_out/dvol_pmc_<tag>.txt
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU \
  --kernel-trace --output-format csv -d gpurun_out/dvolpmc_$tag -- python3 scripts/dvol_micro.py > gpurun_out/dvol_pmc_$tag.log 2>&1
python3 - gpurun_out/dvolpmc_$tag > gpurun_out/dvol_pmc_$tag.txt <<'PY'
import csv, sys, glob, collections
d = sys.argv[1]
ctr = list(csv.DictReader(open(glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0])))
trc = list(csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])))
dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in trc}
per = collections.defaultdict(dict)
for r in ctr:
    per[(r["Dispatch_Id"], r["Kernel_Name"])][r["Counter_Name"]] = float(r["Counter_Value"])
agg = collections.defaultdict(list)
for (did, name), c in per.items():
    if "corr_dvol" not in name or did not in dur:
        continue
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
    agg[(short, int(c.get("SQ_WAVES", 0)))].append((dur[did], c))
print("# PMC per dispatch (median-duration dispatch of each (kernel, wave count)); SQ_* cycle counters are quad-cycles summed over waves")
for (name, waves), v in sorted(agg.items()):
    v.sort(key=lambda x: x[0])
    ns, c = v[len(v) // 2]
    print(f"{name}  waves {waves}  {ns / 1e3:8.1f} us  ({len(v)} dispatches)")
    for k in sorted(c):
        print(f"    {k:24s} {c[k]:.4g}" + (f"   per wave {c[k] / max(waves, 1):.4g}" if k != "SQ_WAVES" else ""))
PY
find gpurun_out/dvolpmc_$tag -name "*.csv" -delete; find gpurun_out/dvolpmc_$tag -name "*.db" -delete
cat gpurun_out/dvol_pmc_$tag.txt | head -80
