# would batching a layer over the 12 iterations of a step pay?  per-launch time at B and at 12 B (graph replay)
for bhw in 4,55,128 48,55,128 2,46,96 24,46,96 1,47,156 12,47,156; do
  echo "=== $bhw"
  CONV_MICRO_GRAPH=1 CONV_MICRO_BHW=$bhw python scripts/conv_micro.py 20 "dg" 2>&1 | grep " fwd " | grep -E "dg cv|dg c2|dg c1|dg f2|dg m2"
done
