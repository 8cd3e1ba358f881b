# same-box A/B of the cache policy of the tiled lookup's window loads (FSRAFT_LOOKUP_POLICY: 0 plain, 2 nt, 16 sc1, 18 both)
for p in 0 2 16 18 -1; do
  FSRAFT_LOOKUP_POLICY=$p python bench.py --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('policy=$p', round(d['value'],2), 'ms', round(d['ms_per_step'],3), 'corr', round(d['roofline_corr']['frac'],4), {n:round(k[n]['ms_per_step'],3) for n in ('corr_build','corr_lookup_fwd','corr_lookup_bwd','corr_build_bwd','conv_igemm')})"
done
