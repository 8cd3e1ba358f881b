mkdir -p gpurun_out/r06
for v in gma alt l2l gma_l2l dropin; do
  extra=""
  [ $v = alt ] && extra="--height 376 --width 1248 --batch-per-gpu 1"
  [ $v = l2l ] && extra="--batch-per-gpu 1"
  [ $v = gma_l2l ] && extra="--batch-per-gpu 1"
  python bench.py --variant $v $extra --no-cpu-baseline > gpurun_out/r06/bench_$v.json 2> gpurun_out/r06/bench_$v.err
  python - <<EOF2
import json
try:
    d=json.loads(open("gpurun_out/r06/bench_$v.json").read().strip().splitlines()[-1])
    print("$v", round(d["value"],2), round(d["ms_per_step"],2), {k:round(v["ms_per_step"],2) for k,v in d.get("kernels",{}).items() if v["ms_per_step"]>0.4})
except Exception as e:
    print("$v FAILED", e)
EOF2
done
python bench.py --height 368 --width 496 --batch-per-gpu 8 --no-cpu-baseline > gpurun_out/r06/bench_chairs.json 2> gpurun_out/r06/bench_chairs.err
python -c "
import json
d=json.loads(open('gpurun_out/r06/bench_chairs.json').read().strip().splitlines()[-1]); print('chairs', round(d['value'],2), round(d['ms_per_step'],2))"
