cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_properties_gpu.py tests/test_misc_ops.py tests/test_backbone_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q 2>&1 | tail -3
timeout 400 python tests/fuzz_ops.py 240 991 2>&1 | tail -1
timeout 300 python tests/fuzz_ops.py 150 17 big 2>&1 | tail -1
