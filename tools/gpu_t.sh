#!/bin/bash
# usage: gpu_t.sh "<pytest args>"   -- build + run a pytest selection on the GPU box
export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
timeout 900 python -m pytest $1 -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -25
