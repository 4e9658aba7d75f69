#!/bin/bash
# round 6: stage table at 128 frames for library variants: bash tools/gpu_r06_var.sh TAG "grep pattern" variant...   ("product" = the in-tree library)
TAG=$1; PAT=$2; shift 2; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
V=$GRAFT_REPO_ROOT/3dssd_amd/csrc/variants
for rep in 1 2; do
for v in "$@"; do
  if [ $v = product ]; then L=""; else L="SA3D_LIB=$V/lib_$v.so"; fi
  echo "== $v (run $rep)"; env $L timeout 300 python tools/stages_at.py 128 2>&1 | grep "$PAT\|total" | tee -a $OUT/stages_$v.txt
done
done
