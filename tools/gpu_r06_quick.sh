cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 200 python tests/fuzz_ops.py 90 4242 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 --extras-budget 0 2>&1 | tail -1 | cut -c1-400
python bench.py --steps 512 --warmup 64 --extras-budget 0 2>&1 | tail -1 | cut -c1-300
python bench.py --steps 512 --warmup 64 --data rings64 --extras-budget 0 2>&1 | tail -1 | cut -c1-300
