#!/bin/bash
# usage: scripts/pmc.sh <tag> "<counters>" <script args...>   -- PMC pass on conv_micro (run from repo root on the GPU box)
tag=$1; shift; ctr=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/pmc_$tag -- python3 scripts/conv_micro.py "$@" > gpurun_out/pmc_$tag.log 2>&1
f=$(find gpurun_out/pmc_$tag -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if 'conv_' in r['Kernel_Name'] or 'gemm' in r['Kernel_Name']:
        agg[r['Kernel_Name'][30:95]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    print(k)
    for c,vals in sorted(v.items()): print('    %-28s %s' % (c, ' '.join('%.4g'%x for x in vals[:3])))
PY
