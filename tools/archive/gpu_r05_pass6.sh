#!/bin/bash
# round 5, GPU pass 6: wide128 with the next item's gather prefetched under the pooling phase -- parity, then A/B against
# the previous kernel (variants/lib_w128old.so) on the single-stream stage timings at 128 frames
OUT=gpurun_out/r05_pass6; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q -x -k "mlp or backbone or pipeline or wide" -p no:cacheprovider -rf > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
for d in default rings64; do
  for v in new old new old; do
    if [ $v = old ]; then export SA3D_LIB=$GRAFT_REPO_ROOT/3dssd_amd/csrc/variants/lib_w128old.so; else unset SA3D_LIB; fi
    echo "== data=$d wide128=$v"; timeout 300 python tools/stages_at.py 128 data=$d 2>&1 | grep -v amdgpu.ids | grep "m=256\|total"
  done
done
unset SA3D_LIB
echo "== done"
