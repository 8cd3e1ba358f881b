cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for i in 1 2; do timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -1; done
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c1-220
