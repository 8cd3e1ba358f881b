#!/bin/bash
# A/B of core/streams.py (independent branches of the forward pass on a second HIP stream) on one box: hipGraph-replayed steps of
# bench.py with the switch off / on, twice each.  usage (GPU box, repo root): bash scripts/ab_stream_overlap.sh [variant ...]
what=overlap
for v in "${@:-raft}"; do
  extra=""
  case $v in l2l|gma_l2l) extra="'--batch-per-gpu','1',";; alt) extra="'--height','376','--width','1248','--batch-per-gpu','1',";; esac
  for i in 0 1 0 1; do
    python -c "
import sys
sys.argv=['bench.py','--variant','$v',$extra'--steps','20','--warmup','3','--no-cpu-baseline','--no-extra','--no-kernel-timing']
from flow_supervisor_amd.core import streams
streams.OVERLAP=bool($i)
import bench
bench.main()
" 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$v $what=$i', round(d['value'],2), 'pairs/s', round(d['ms_per_step'],3), 'ms')"
  done
done
