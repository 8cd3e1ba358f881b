# usage (GPU box, repo root): bash scripts/probes/word_cost_layers.sh  -- conv_micro layers with the amax words read + raised / read only / absent
mkdir -p gpurun_out/r06
for i in 1 2; do
CONV_MICRO_GRAPH=1 python scripts/conv_micro.py 20 > gpurun_out/r06/micro4_full_$i.txt 2>&1
FSRAFT_NO_WORDS=2 CONV_MICRO_GRAPH=1 python scripts/conv_micro.py 20 > gpurun_out/r06/micro4_noraise_$i.txt 2>&1
FSRAFT_NO_WORDS=3 CONV_MICRO_GRAPH=1 python scripts/conv_micro.py 20 > gpurun_out/r06/micro4_nowords_$i.txt 2>&1
done
