#!/bin/bash
# round 6: the sorting form of the grid ball query -- parity (tests + fuzz), then the stage timings on three kinds of frames, A/B against the list form
# (variant library: tools/build_variant.sh bqlist ballquery_grid "-DSA_BQ_FORCE_LIST=1")
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06_bqsort; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
if [ "$1" != "noparity" ]; then
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_properties_gpu.py tests/test_misc_ops.py -m gpu -x -q -k "ball or query or grid or group" 2>&1 | tail -8 > $OUT/pytest.txt; cat $OUT/pytest.txt
timeout 300 python tests/fuzz_ops.py 120 777 > $OUT/fuzz_small.txt 2>&1; tail -2 $OUT/fuzz_small.txt
fi
for d in default rings64 dense; do
  f=128; [ $d = dense ] && f=32
  python tools/stages_at.py $f data=$d 2>&1 | grep "ball_query\|non-FPS" > $OUT/stages_${d}_sort.txt
  SA3D_LIB=$PWD/3dssd_amd/csrc/variants/lib_bqlist.so python tools/stages_at.py $f data=$d 2>&1 | grep "ball_query\|non-FPS" > $OUT/stages_${d}_list.txt
  echo "== $d sort"; cat $OUT/stages_${d}_sort.txt;  echo "== $d list"; cat $OUT/stages_${d}_list.txt
done
