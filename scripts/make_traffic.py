#!/usr/bin/env python3
"""profiles/traffic.json (per-launch HBM bytes per kernel family, read by bench.py for `roofline.traffic`) from the raw PMC
file that scripts/pmc_traffic.sh writes.  usage: python scripts/make_traffic.py gpurun_out/traffic_<tag>.json

Counter handling (MI355X_MICROARCH.md, HBM section): on gfx950 FETCH_SIZE reports half the bytes of a read stream issued as
>= 8-byte-per-lane loads, and EVERY kernel family listed here loads 16 bytes per lane (GEMM staging, LDS-DMA pieces, the
lookups' tile rows, float4 epilogues), so FETCH_SIZE is doubled for all of them; WRITE_SIZE is exact for 16-byte stores and
float atomics.  (Round 1 left the factor at 1 for the correlation kernels on the wrong assumption that they issue 4-byte
gathers; VERDICT r1, weak #4.)  The second output column is traffic / algorithmic bytes where the family has a byte model."""
import json
import sys

raw = json.load(open(sys.argv[1]))
FAM = [("conv_igemm", "conv_igemm"), ("conv3x3_halo", "conv_igemm"), ("conv_rec", "conv_igemm"), ("conv_patch", "conv_igemm"), ("conv_wgrad", "conv_wgrad"),
       ("gemm_split", "gemm_f32"), ("gemm_kernel", "gemm_f32"), ("gemm_rec", "gemm_f32"),
       ("corr_build_rec", "corr_build"), ("corr_build_tiled", "corr_build"), ("corr_build_split", "corr_build"),
       ("lookup_tiled_fwd", "corr_lookup_fwd"), ("corr_lookup_fwd", "corr_lookup_fwd"),
       ("corr_dvol", "corr_lookup_bwd"), ("corr_lookup_bwd", "corr_lookup_bwd"),
       ("altcorr_fused_fwd", "altcorr_fwd"), ("upsample_fwd", "upsample_fwd")]
acc = {}
for k, v in raw.items():
    for pre, fam in FAM:
        if k.startswith(pre):
            b = (v["fetch_KiB_avg"] * 2.0 + v["write_KiB_avg"]) * 1024.0 * v["launches"]
            a = acc.setdefault(fam, [0.0, 0])
            a[0] += b
            # the gradient volume of a step: one timed launch (ops.corr_dvol_build) = the bounding-box / separable kernel plus the
            # row kernel's (usually empty) walks over its work list -- bytes per ops-level launch, like bench.py's avg_launch_us
            if fam != "corr_lookup_bwd" or not k.startswith("corr_dvol_kernel"):
                a[1] += v["launches"]
            break
out = {fam: a[0] / max(a[1], 1) for fam, a in acc.items()}
# provenance: the commit and run the counters came from (bench.py copies it into the line as traffic_source)
import os
# ... and the workload they were counted on: bench.py only attaches these bytes to a line of the SAME (variant, height, width, pairs
# per GPU, iterations) and prints `traffic: null` otherwise (VERDICT r4 weak #8a).  FSRAFT_TRAFFIC_KEY="variant,H,W,batch,iters"
kv = os.environ.get("FSRAFT_TRAFFIC_KEY", "raft,440,1024,4,12").split(",")
out["_meta"] = {"commit": os.environ.get("FSRAFT_COMMIT", "unknown"), "raw": os.path.basename(sys.argv[1]),
                "collected": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, one eager bench step (scripts/pmc_traffic.sh)",
                "key": {"variant": kv[0], "height": int(kv[1]), "width": int(kv[2]), "batch_per_gpu": int(kv[3]), "iters": int(kv[4])}}
json.dump(out, open("profiles/traffic.json", "w"), indent=1)
# algorithmic bytes per launch at the bench shape (4 pairs, 55x128, C=256, r=4, 12 lookups)
N, P, C, B = 7040, 9280, 256, 4
ALG = {"corr_build": 4.0 * B * (2 * N * C + N * P), "corr_lookup_fwd": 4.0 * B * N * (400 + 2 + 324),
       "corr_lookup_bwd": 4.0 * B * N * (12 * (324 + 2 + 800) + P) / 2.0}
for k, v in out.items():
    if k.startswith("_"):
        continue
    r = f"   traffic / algorithmic = {v / ALG[k]:.2f}" if k in ALG else ""
    print(f"{k:18s} {v / 1e6:9.1f} MB per launch{r}")
