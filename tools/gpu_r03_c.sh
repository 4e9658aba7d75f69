#!/bin/bash
TAG=${1:-r03c}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests -m gpu -q -x -k "group_mlp or backbone or pipeline or head or properties" -p no:cacheprovider -rf > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
bash tools/gpu_variants.sh $TAG base
