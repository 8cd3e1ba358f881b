for e in 1 2; do
  FSRAFT_TN_LIST_KSPLIT=$e python bench.py --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('tn ksplit=$e', round(d['value'],2), 'ms', round(d['ms_per_step'],3), 'corr', round(d['roofline_corr']['frac'],4), {n:round(k[n]['ms_per_step'],3) for n in ('corr_build','corr_lookup_fwd','corr_lookup_bwd','gemm_f32','corr_build_bwd')})"
done
bash scripts/prof.sh r03k > /dev/null 2>&1
grep -i "gemm_rec\|ktiles\|f2cat\|to_records\|dfmap2\|zero_matri\|dvol\|corr_build\|lookup_tiled" gpurun_out/kernel_stats_r03k.csv | cut -d, -f1-4 | cut -c1-150
