mkdir -p gpurun_out/r06
python scripts/conv_accuracy.py > gpurun_out/r06/accuracy_scaled.txt 2>&1
for i in 1 2; do
CONV_MICRO_GRAPH=1 python scripts/conv_micro.py 20 > gpurun_out/r06/micro3_scaled_$i.txt 2>&1
FSRAFT_LIB_PATH=$PWD/flow_supervisor_amd/libfsraft_bf16.so CONV_MICRO_GRAPH=1 python scripts/conv_micro.py 20 > gpurun_out/r06/micro3_bf16_$i.txt 2>&1
done
tail -n 20 gpurun_out/r06/accuracy_scaled.txt
