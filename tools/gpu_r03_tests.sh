#!/bin/bash
TAG=${1:-r03t}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider -rf > $OUT/pytest_gpu.log 2>&1; tail -6 $OUT/pytest_gpu.log
bash tools/gpu_variants.sh $TAG base
