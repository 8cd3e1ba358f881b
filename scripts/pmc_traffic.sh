#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes as MI355X_MICROARCH.md prescribes) of every kernel of one
# bench step.  Run from the repo root on the GPU box; writes gpurun_out/traffic_<tag>.json
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MIOPEN_FIND_MODE=2
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmct_${tag}_$c -- python3 bench.py --steps 1 --warmup 1 --graph 0 --one-stream --no-cpu-baseline --no-kernel-timing --no-extra > gpurun_out/pmct_${tag}_$c.log 2>&1
done
python3 - "$tag" <<'PY'
import csv, glob, json, sys, collections
tag = sys.argv[1]
out = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmct_{tag}_{c}/*/*counter_collection.csv")[0]
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "anonymous namespace" not in n or "at::native" in n:
            continue
        key = n.split("(anonymous namespace)::")[1].split("(")[0][:70]
        out[key][c].append(float(r["Counter_Value"]))
res = {}
for k, v in out.items():
    # counters are in KiB; keep the second half of the launches (the timed step, after warm-up)
    fs, ws = v.get("FETCH_SIZE", [0]), v.get("WRITE_SIZE", [0])
    fs, ws = fs[len(fs) // 2:], ws[len(ws) // 2:]
    res[k] = {"launches": len(fs), "fetch_KiB_avg": sum(fs) / max(len(fs), 1), "write_KiB_avg": sum(ws) / max(len(ws), 1)}
json.dump(res, open(f"gpurun_out/traffic_{tag}.json", "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["fetch_KiB_avg"] * kv[1]["launches"]):
    print(f"{k:72s} n={v['launches']:4d} fetch {v['fetch_KiB_avg']/1024:9.1f} MiB  write {v['write_KiB_avg']/1024:9.1f} MiB")
PY
rm -rf gpurun_out/pmct_${tag}_FETCH_SIZE gpurun_out/pmct_${tag}_WRITE_SIZE
