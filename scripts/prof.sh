#!/bin/bash
# usage: scripts/prof.sh <tag> [extra bench.py args]   (runs on the GPU box from the repo root)
tag=$1
shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MIOPEN_FIND_MODE=2
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --steps 2 --warmup 1 --graph 0 --one-stream --no-cpu-baseline --no-kernel-timing --no-extra "$@" > gpurun_out/prof_$tag.log 2>&1
tail -1 gpurun_out/prof_$tag.log | cut -c1-400
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/kernel_stats_$tag.csv
find gpurun_out/prof_$tag -name "*kernel_trace.csv" -delete; find gpurun_out/prof_$tag -name "*.db" -delete
