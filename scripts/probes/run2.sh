mkdir -p gpurun_out/r06
./scripts/probes/split_mix.bin > gpurun_out/r06/split_mix.txt 2>&1
python scripts/conv_accuracy.py > gpurun_out/r06/accuracy_f16mix_s1.txt 2>&1
for i in 1 2; do
for v in mix vec bf16; do
lib=$PWD/flow_supervisor_amd/libfsraft_$v.so; [ $v = mix ] && lib=$PWD/flow_supervisor_amd/libfsraft.so
FSRAFT_LIB_PATH=$lib CONV_MICRO_GRAPH=1 python scripts/conv_micro.py 20 > gpurun_out/r06/micro2_${v}_$i.txt 2>&1
done
done
cat gpurun_out/r06/split_mix.txt
