#!/bin/bash
# quick GPU check: the MLP / backbone / pipeline tests, then single-stream stage timings of library variants ($@, "base" = product)
TAG=${TAG:-quick}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests -m gpu -q -x -k "${KEXPR:-group_mlp or backbone or pipeline}" -p no:cacheprovider -rf > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
bash tools/gpu_variants.sh $TAG "$@"
