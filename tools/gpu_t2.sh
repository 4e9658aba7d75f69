#!/bin/bash
export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_backbone_gpu.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -8
timeout 300 python tools/bench_ops.py 2>&1 | grep -v amdgpu.ids | grep "query_ball\|configs" 
