# usage (GPU box, repo root): bash scripts/probes/amax_trace_variants.sh  -- FSRAFT_AMAX_TRACE per bench variant: call sites that still run a standalone amax pass
for v in gma alt l2l; do
  extra=""
  [ $v = alt ] && extra="--height 376 --width 1248 --batch-per-gpu 1"
  [ $v = l2l ] && extra="--batch-per-gpu 1"
  echo "== $v"
  FSRAFT_AMAX_TRACE=1 python bench.py --variant $v $extra --steps 1 --warmup 1 --graph 0 --no-cpu-baseline --no-kernel-timing --no-extra 2>&1 | grep "amax pass" | head -24
done
