export MIOPEN_FIND_MODE=2
timeout 600 python -m pytest tests -q -m gpu -k "encoder_conv or channels_last" 2>&1 | grep -v Warning | grep -E "^E  |passed|failed|^FAILED" | head -8
for x in 0 1; do echo "xcd=$x"; FSRAFT_WGRAD_XCD=$x timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k:(round(v['ms_per_step'],2)) for k,v in d.get('kernels',{}).items() if 'wgrad' in k})"; 
CONV_MICRO_BHW=8,220,512 timeout 120 python scripts/conv_micro.py 5 e1 22=$x 2>&1 | grep wgrad; done
