for bhw in 2,54,128 2,46,96 4,55,128; do
  echo "=== $bhw graph"
  CONV_MICRO_GRAPH=1 CONV_MICRO_BHW=$bhw python scripts/conv_micro.py 50 - 2>&1 | grep " fwd "
done
echo "=== 2,54,128 eager"
CONV_MICRO_BHW=2,54,128 python scripts/conv_micro.py 50 y 2>&1 | grep " fwd "
