for e in 0 1 2; do
  FSRAFT_KTILE_EXACT=$e FSRAFT_KTILE_STATS=1 python bench.py --steps 2 --warmup 1 --graph 0 --no-cpu-baseline --no-extra --no-kernel-timing 2>&1 | grep "k-tiles kept" | tail -1
  FSRAFT_KTILE_EXACT=$e python bench.py --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('exact=$e', round(d['value'],2), 'ms', round(d['ms_per_step'],3), 'corr', round(d['roofline_corr']['frac'],4), {n:round(k[n]['ms_per_step'],3) for n in ('corr_build','corr_lookup_fwd','corr_lookup_bwd','gemm_f32','corr_build_bwd')})"
done
FSRAFT_KTILE_EXACT=2 timeout 300 python -m pytest tests -m gpu -q -x -s -k "listed_k_tiles" 2>&1 | tail -8
