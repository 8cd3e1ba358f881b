# refresh of the judged evidence: bench line via the driver's launch line, kernel stats, PMC traffic, variants, round-4 micro evidence
# usage (on the GPU box, repo root): FSRAFT_COMMIT=<short sha> bash scripts/refresh_evidence.sh <tag>   -> gpurun_out/*_<tag>*,
# copied to profiles/ by hand (gpurun_out/ is scratch)
tag=${1:-r06}
export MIOPEN_FIND_MODE=2
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/bench_${tag}_1gpu.json
cut -c1-300 gpurun_out/bench_${tag}_1gpu.json
bash scripts/prof.sh ${tag}
bash scripts/pmc_traffic.sh ${tag} | head -14
python scripts/make_traffic.py gpurun_out/traffic_${tag}.json > gpurun_out/traffic_${tag}.txt 2>&1; cp profiles/traffic.json gpurun_out/traffic_${tag}_families.json
# the line again with this run's traffic.json in place (roofline.traffic, hbm_gbs_counters, traffic_source)
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/bench_${tag}_1gpu.json
# python bench.py --gpus 2 typed as is: the ranks share the box's GPU and exchange over gloo through the host
python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_${tag}_gpus2_shared.json; cut -c1-200 gpurun_out/bench_${tag}_gpus2_shared.json
python bench.py --variant gma --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_${tag}_gma.json; cut -c1-200 gpurun_out/bench_${tag}_gma.json
python bench.py --variant alt --height 376 --width 1248 --batch-per-gpu 1 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_${tag}_alt.json; cut -c1-200 gpurun_out/bench_${tag}_alt.json
# the alt variant's kernel summary (no r04 summary was kept) and the reference-shaped shell over the swapped blocks
bash scripts/prof.sh ${tag}_alt --variant alt --height 376 --width 1248 --batch-per-gpu 1
python bench.py --variant dropin --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_${tag}_dropin.json; cut -c1-200 gpurun_out/bench_${tag}_dropin.json
python bench.py --height 368 --width 496 --batch-per-gpu 8 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_${tag}_chairs.json; cut -c1-200 gpurun_out/bench_${tag}_chairs.json
python bench.py --variant l2l --batch-per-gpu 1 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_${tag}_l2l.json; cut -c1-300 gpurun_out/bench_${tag}_l2l.json
python bench.py --variant gma_l2l --batch-per-gpu 1 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_${tag}_gma_l2l.json; cut -c1-300 gpurun_out/bench_${tag}_gma_l2l.json
# round 6: flow regimes: the headline and the alt variant from a rough warm start, the alt lookup's dispatch sweep
python bench.py --flow-regime rough --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_${tag}_rough.json; cut -c1-200 gpurun_out/bench_${tag}_rough.json
python bench.py --variant alt --height 376 --width 1248 --batch-per-gpu 1 --no-cpu-baseline --flow-regime rough 2>&1 | tail -1 > gpurun_out/bench_${tag}_alt_rough.json; cut -c1-200 gpurun_out/bench_${tag}_alt_rough.json
FSRAFT_ALT_DISPATCH=0 python bench.py --variant alt --height 376 --width 1248 --batch-per-gpu 1 --no-cpu-baseline --flow-regime rough 2>&1 | tail -1 > gpurun_out/bench_${tag}_alt_rough_nodispatch.json; cut -c1-200 gpurun_out/bench_${tag}_alt_rough_nodispatch.json
python scripts/alt_mfma_micro.py 2>&1 | tail -19 > gpurun_out/altcorr_regimes_${tag}.txt
# round 6: the split arithmetic against fp64 per layer shape; stall counters of the convolution kernels
python scripts/conv_accuracy.py > gpurun_out/conv_accuracy_${tag}.txt 2>&1
python3 scripts/stall_pmc.py ${tag} c2 hd zrc > /dev/null 2>&1
bash scripts/prof.sh ${tag}_dropin --variant dropin
# gradient-volume kernels under PMC, the lookup's gather floor, per-layer convolution times
bash scripts/dvol_pmc.sh ${tag} > /dev/null 2>&1
python scripts/dvol_micro.py 2>&1 | tail -9 > gpurun_out/dvol_micro_${tag}.txt
python scripts/lookup_gather_floor.py 2>&1 | tail -2 > gpurun_out/lookup_gather_floor_${tag}.txt
python scripts/layer_times.py > gpurun_out/layer_times_${tag}.txt 2>&1
# matrix-pipe busy x clock of the convolution kernels (PMC + kernel trace)
{
  echo "# Matrix-pipe busy fraction and the clock the convolution kernels ran at (scripts/mfma_busy.sh: one PMC pass with the kernel trace,"
  echo "# SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) and GRBM_GUI_ACTIVE / 8 / duration), commit ${FSRAFT_COMMIT:-?}."
  for l in c2 zrc qc c1 hd; do
    echo "== layer $l (scripts/conv_micro.py, 4 x 55 x 128; forward + 12-segment weight gradient)"
    bash scripts/mfma_busy.sh ${tag}_$l $l
  done
  for cfg in "e1 8,220,512" "e2 8,110,256" "e3 8,55,128"; do
    set -- $cfg
    echo "== layer $1 at B,H,W = $2 (one weight-gradient segment)"
    CONV_MICRO_BHW=$2 CONV_MICRO_NSEG=1 bash scripts/mfma_busy.sh ${tag}_$1 $1
  done
} > gpurun_out/mfma_busy_clock_${tag}.txt 2>&1
# the second-stream routes off / on (core/streams.py), hipGraph replays on this box
bash scripts/ab_stream_overlap.sh raft gma l2l gma_l2l alt > gpurun_out/stream_overlap_${tag}.txt 2>&1
