#!/bin/bash
# Engine clock while a kernel family runs in a loop: MFMA-only (short-K style loop is not in the tree any more, so: the
# exact-fp32 and the split-bf16 gate convolution, and an HBM-bound norm kernel) -- is the split core running at 2.4 GHz?
# usage (GPU box, repo root): bash scripts/clock_probe.sh
run() {   # $1 = label, rest = command
  label=$1; shift
  "$@" > /dev/null 2>&1 &
  pid=$!
  sleep 14
  echo "== $label"
  for i in 1 2 3; do rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -2 | tr -s ' ' | cut -c1-100; sleep 1; done
  rocm-smi --showpower 2>/dev/null | grep -i "power" | head -2 | cut -c1-100
  wait $pid
}
run "idle" sleep 16
run "split-bf16 1x5 256->256 convolution, 400000 launches" python scripts/conv_micro.py 400000 zrc
run "exact-fp32 same layer, 80000 launches" python scripts/conv_micro.py 80000 zrc 3=0
run "whole train step (bench.py, 400 steps)" python bench.py --steps 400 --warmup 3 --no-cpu-baseline --no-extra --no-kernel-timing
