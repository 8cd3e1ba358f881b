# usage (GPU box, repo root): bash scripts/probes/word_cost_wgrad.sh  -- 12-segment weight gradients with distinct / shared / no amax words
mkdir -p gpurun_out/r06
for i in 1 2; do
for l in "c2 " "zr " "q  " "hd " "m2 " "c1 " "cv "; do
CONV_MICRO_NSEG=12 CONV_MICRO_GRAPH=1 python scripts/conv_micro.py 10 "$l" 2>&1 | grep wgrad | sed "s/^/distinct /"
CONV_MICRO_SHAREWORD=1 CONV_MICRO_NSEG=12 CONV_MICRO_GRAPH=1 python scripts/conv_micro.py 10 "$l" 2>&1 | grep wgrad | sed "s/^/shared   /"
FSRAFT_NO_WORDS=3 CONV_MICRO_NSEG=12 CONV_MICRO_GRAPH=1 python scripts/conv_micro.py 10 "$l" 2>&1 | grep wgrad | sed "s/^/nowords  /"
done
done > gpurun_out/r06/wgrad_words.txt 2>&1
cat gpurun_out/r06/wgrad_words.txt
