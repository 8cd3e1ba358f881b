# k-slices of the two list GEMMs of the volume backward (same-box A/B)
for cfg in "1 2" "2 2" "1 3" "2 3" "1 4"; do
  set -- $cfg
  FSRAFT_NT_LIST_KSPLIT=$1 FSRAFT_TN_LIST_KSPLIT=$2 python bench.py --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('nt=$1 tn=$2', round(d['ms_per_step'],3), 'corr', round(d['roofline_corr']['frac'],4), {n:round(k[n]['ms_per_step'],3) for n in ('gemm_f32','corr_build_bwd')})"
done
