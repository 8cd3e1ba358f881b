import json, sys
j = json.load(open(sys.argv[1]))
print(j["value"], "pairs/s", j["ms_per_step"], "ms/step")
for k, x in j.get("kernels", {}).items():
    print(f"   {k:18s} {x['ms_per_step']:8.2f} ms/step  avg {x['avg_launch_us']:8.1f} us  {x['achieved']:8.1f} {x['unit']} frac {x['frac']:.3f}")
