export MIOPEN_FIND_MODE=2
timeout 900 python -m pytest tests -q -m gpu 2>&1 | grep -v Warning | grep -E "^E  |passed|failed|^FAILED" | head -8
timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300
