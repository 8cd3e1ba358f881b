# refresh of the judged evidence: bench line via the driver's launch line, kernel stats, PMC traffic, variants
export MIOPEN_FIND_MODE=2
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 2>&1 | tail -1 > gpurun_out/bench_final.json
cut -c1-400 gpurun_out/bench_final.json
bash scripts/prof.sh r1n
bash scripts/pmc_traffic.sh r1n | head -12
python bench.py --variant gma --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_gma.json; cut -c1-200 gpurun_out/bench_gma.json
python bench.py --variant alt --height 376 --width 1248 --batch-per-gpu 1 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_alt.json; cut -c1-200 gpurun_out/bench_alt.json
python bench.py --height 368 --width 496 --batch-per-gpu 8 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_chairs.json; cut -c1-200 gpurun_out/bench_chairs.json
