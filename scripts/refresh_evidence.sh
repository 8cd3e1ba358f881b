# refresh of the judged evidence: bench line via the driver's launch line, kernel stats, PMC traffic, variants
# usage (on the GPU box, repo root): bash scripts/refresh_evidence.sh <tag>      -> gpurun_out/*_<tag>*, copied to profiles/ by hand
tag=${1:-r03}
export MIOPEN_FIND_MODE=2
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/bench_${tag}_1gpu.json
cut -c1-300 gpurun_out/bench_${tag}_1gpu.json
bash scripts/prof.sh ${tag}
bash scripts/pmc_traffic.sh ${tag} | head -14
python bench.py --variant gma --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_${tag}_gma.json; cut -c1-200 gpurun_out/bench_${tag}_gma.json
bash scripts/prof.sh ${tag}_gma --variant gma
python bench.py --variant alt --height 376 --width 1248 --batch-per-gpu 1 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_${tag}_alt.json; cut -c1-200 gpurun_out/bench_${tag}_alt.json
bash scripts/prof.sh ${tag}_alt --variant alt --height 376 --width 1248 --batch-per-gpu 1
python bench.py --height 368 --width 496 --batch-per-gpu 8 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_${tag}_chairs.json; cut -c1-200 gpurun_out/bench_${tag}_chairs.json
python bench.py --variant l2l --batch-per-gpu 1 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_${tag}_l2l.json; cut -c1-300 gpurun_out/bench_${tag}_l2l.json
bash scripts/prof.sh ${tag}_l2l --variant l2l --batch-per-gpu 1
python bench.py --variant gma_l2l --batch-per-gpu 1 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_${tag}_gma_l2l.json; cut -c1-300 gpurun_out/bench_${tag}_gma_l2l.json
