export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "square_dist" -p no:cacheprovider 2>&1 | tail -3
timeout 120 python tools/sqdist_prof.py 32 128
