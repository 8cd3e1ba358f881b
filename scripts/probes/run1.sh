set -x
mkdir -p gpurun_out/r06
./scripts/probes/denorm.bin > gpurun_out/r06/denorm.txt 2>&1
python scripts/conv_accuracy.py > gpurun_out/r06/accuracy_f16_s1.txt 2>&1
FSRAFT_LIB_PATH=$PWD/flow_supervisor_amd/libfsraft_bf16.so python scripts/conv_accuracy.py > gpurun_out/r06/accuracy_bf16.txt 2>&1
for i in 1 2; do
CONV_MICRO_GRAPH=1 python scripts/conv_micro.py 20 > gpurun_out/r06/micro_f16_$i.txt 2>&1
FSRAFT_LIB_PATH=$PWD/flow_supervisor_amd/libfsraft_bf16.so CONV_MICRO_GRAPH=1 python scripts/conv_micro.py 20 > gpurun_out/r06/micro_bf16_$i.txt 2>&1
done
cat gpurun_out/r06/denorm.txt
