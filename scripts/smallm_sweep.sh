# small-M sweep of the update-block convolution shapes: which tile configuration / route wins at one or two pairs per GPU
for bhw in 2,54,128 2,46,96 1,47,156; do
  for cfg in "" "13=0" "13=0 32=0" "32=2" "32=3" "26=2 31=2048" "14=100000 5=2"; do
    echo "=== $bhw | $cfg"
    CONV_MICRO_BHW=$bhw python scripts/conv_micro.py 50 - $cfg 2>&1 | grep " fwd " | grep -v "^e[123]\|^x" 
  done
done
