for b in 0 1 0 1; do
  FSRAFT_DVOL_BOX=$b python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null > /tmp/ab_$b.json
  python - <<PY
import json
d = json.load(open("/tmp/ab_$b.json")); k = d["kernels"]
print("box $b", round(d["value"], 2), round(d["ms_per_step"], 3), "lookup_bwd ms", round(k["corr_lookup_bwd"]["ms_per_step"], 3), "corr frac", round(d["roofline_corr"]["frac"], 3))
PY
done
