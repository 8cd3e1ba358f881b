#!/usr/bin/env python3
"""Where the SIMD cycles of a convolution kernel go (VERDICT r5 next #2 iii): issue / wait / busy counters of the SQ block for
the kernels of one scripts/conv_micro.py layer, collected in separate PMC passes of eight SQ counters each (MI355X_MICROARCH.md,
"rocprofv3 PMC slots") and printed as fractions of SQ_WAVE_CYCLES.

usage (GPU box, repo root):  python3 scripts/stall_pmc.py <tag> <layer> [<layer> ...]      -> gpurun_out/stall_pmc_<tag>.txt
This process never touches the GPU itself; rocprofv3 runs conv_micro.py as its own child.
"""
import collections
import csv
import glob
import os
import shutil
import subprocess
import sys

PASSES = [
    ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_WAVES", "GRBM_GUI_ACTIVE"],
    ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_FLAT",
     "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INST_CYCLES_VMEM"],
    ["SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"],
    ["SQ_VALU_MFMA_COEXEC_CYCLES", "SQ_INST_LEVEL_VMEM", "SQ_INST_LEVEL_LDS", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_UNALIGNED_STALL", "SQ_WAIT_INST_VALU",
     "SQ_INSTS_VALU_MFMA_MOPS_F16", "SQ_WAVE32_INSTS"],
]


def available():
    out = subprocess.run(["rocprofv3", "-L"], capture_output=True, text=True).stdout
    names = set()
    for tok in out.replace(",", " ").replace(":", " ").split():
        if tok.startswith(("SQ_", "GRBM_", "TCC_", "TCP_")):
            names.add(tok.strip())
    return names


def one_pass(tag, i, ctrs, layer):
    d = f"gpurun_out/stall_{tag}_{layer}_{i}"
    shutil.rmtree(d, ignore_errors=True)
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--pmc", *ctrs, "--output-format", "csv", "-d", os.path.abspath(d), "--", "python3",
           os.path.abspath("scripts/conv_micro.py"), "3", layer]
    r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True)
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not files:
        return None, (r.stdout + r.stderr)[-400:]
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(files[0])):
        name = row["Kernel_Name"]
        if "conv_" not in name and "conv3x3" not in name:
            continue
        short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:80]
        per[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
    shutil.rmtree(d, ignore_errors=True)
    return per, ""


def main():
    tag, layers = sys.argv[1], sys.argv[2:]
    os.chdir(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
    have = available()
    lines = [f"# stall-reason counters, scripts/stall_pmc.py (commit {os.environ.get('FSRAFT_COMMIT', '?')}); last launch of each kernel of the layer;",
             "# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; WAIT_ANY (parked on s_waitcnt / barrier) +",
             "# WAIT_INST_ANY (ready but not issued) + ACTIVE_INST_ANY (issuing) ~ WAVE_CYCLES (MI355X_MICROARCH.md, PMC slots)"]
    for layer in layers:
        merged = collections.defaultdict(dict)
        for i, want in enumerate(PASSES):
            ctrs = [c for c in want if not have or c in have]
            missing = [c for c in want if c not in ctrs]
            if missing:
                lines.append(f"# layer {layer} pass {i}: not on this part: {' '.join(missing)}")
            if not ctrs:
                continue
            per, err = one_pass(tag, i, ctrs, layer)
            if per is None:
                lines.append(f"# layer {layer} pass {i} failed: {err!r}")
                continue
            for k, c in per.items():
                for n, v in c.items():
                    merged[k][n] = v[-1]
        for k, c in merged.items():
            lines.append(f"== layer {layer}: {k}")
            wc = c.get("SQ_WAVE_CYCLES")
            for n in sorted(c):
                frac = f"   {c[n] / wc:7.3f} of SQ_WAVE_CYCLES" if wc and n.startswith(("SQ_WAIT", "SQ_ACTIVE", "SQ_INST_CYCLES", "SQ_INST_LEVEL")) else ""
                lines.append(f"    {n:34s} {c[n]:14.5g}{frac}")
            if wc and "SQ_WAVES" in c:
                lines.append(f"    quad-cycles per wave {wc / c['SQ_WAVES']:.0f}")
            if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
                lines.append(f"    LDS bank-conflict cycles / LDS active cycles = {c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']:.3f}")
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("GRBM_GUI_ACTIVE"):
                lines.append(f"    matrix pipe busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * c['GRBM_GUI_ACTIVE'] / 8):.3f} of the SIMD cycles")
    txt = "\n".join(lines) + "\n"
    open(f"gpurun_out/stall_pmc_{tag}.txt", "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
