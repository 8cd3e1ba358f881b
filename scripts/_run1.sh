export MIOPEN_FIND_MODE=2
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "resident_patch" 2>&1 | grep -v Warning | grep -E "^E  |passed|failed|^FAILED" | head -12
CONV_MICRO_BHW=8,220,512 timeout 120 python scripts/conv_micro.py 5 e1 2>&1 | grep fwd
export CONV_MICRO_BHW=8,220,512
bash scripts/pmc.sh h2 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" 2 e1 > /dev/null 2>&1
