#!/bin/bash
# Matrix-pipe busy fraction and effective shader clock of the convolution kernels, from one PMC pass with the kernel trace:
#   busy  = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8)      clock = GRBM_GUI_ACTIVE / 8 / kernel duration
#   (GRBM_GUI_ACTIVE comes back summed over the 8 XCDs)
# usage (GPU box, repo root): scripts/mfma_busy.sh <tag> <conv_micro layer> [key=value ...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export CONV_MICRO_NSEG=${CONV_MICRO_NSEG:-12}
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA --kernel-trace --output-format csv -d gpurun_out/busy_$tag -- python3 scripts/conv_micro.py 3 "$@" > gpurun_out/busy_$tag.log 2>&1
python3 - gpurun_out/busy_$tag <<'PY'
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
    if ("conv_" not in name and "conv3x3" not in name) or did not in dur or "GRBM_GUI_ACTIVE" not in c:
        continue
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
    agg[short].append((dur[did], c))
for name, v in agg.items():
    ns, c = v[-1]
    act = c["GRBM_GUI_ACTIVE"] / 8.0
    print(f"{name}\n    {ns / 1e3:8.1f} us  GRBM_GUI_ACTIVE / 8 = {act:.4g} cycles -> {act / ns:.2f} GHz   MFMA busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * act):.3f} of the SIMD cycles"
          f"   ({c['SQ_INSTS_MFMA']:.4g} MFMAs, {c['SQ_VALU_MFMA_BUSY_CYCLES'] / c['SQ_INSTS_MFMA']:.1f} busy cycles each; at 2.4 GHz the same MFMAs are {c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / 2.4e9 * 1e6:.0f} us)")
PY
find gpurun_out/busy_$tag -name "*.csv" -delete; find gpurun_out/busy_$tag -name "*.db" -delete
