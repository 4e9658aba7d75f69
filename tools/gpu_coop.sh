#!/bin/bash
OUT=gpurun_out/coop; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_properties_gpu.py -x -q -m gpu -k "fps" 2>&1 | tail -3
timeout 300 python tools/fps_coop_bench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/coop_xcd.jsonl
SA_FPS_COOP_XCD=0 timeout 300 python tools/fps_coop_bench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/coop_plainmap.jsonl
