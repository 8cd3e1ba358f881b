timeout 600 python -m pytest tests -m gpu -q -x -s -k "listed_k_tiles or bounding_box or corr or volume or train_step" 2>&1 | tail -12
for k in "1 1" "1 0" "0 0"; do
  set -- $k
  FSRAFT_BWD_KSKIP=$1 FSRAFT_DVOL_WMASK=$2 python bench.py --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('kskip=$1 wmask=$2', round(d['value'],2), 'ms', round(d['ms_per_step'],3), 'corr', round(d['roofline_corr']['frac'],4), {n:round(k[n]['ms_per_step'],3) for n in ('corr_build','corr_lookup_fwd','corr_lookup_bwd','gemm_f32','corr_build_bwd')})"
done
