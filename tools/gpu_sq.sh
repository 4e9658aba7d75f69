#!/bin/bash
export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_backbone_gpu.py tests/test_properties_gpu.py -x -q -m gpu -k "square or backbone or symmetric" 2>&1 | grep -v amdgpu.ids | tail -6
bash tools/gpu_env_ab.sh sq "packed:SA_SQDIST_PACKED=1;v2:SA_SQDIST_PACKED=0" calc_square
