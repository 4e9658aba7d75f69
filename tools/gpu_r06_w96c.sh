#!/bin/bash
# round 6: phase clocks (64 frames) + layer-4 call (128 frames) of the library variants named on the command line
TAG=$1; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
V=$GRAFT_REPO_ROOT/3dssd_amd/csrc/variants
for v in "$@"; do
  if [ -f $V/lib_${v}_prof.so ]; then echo "== phase clocks $v"; SA3D_LIB=$V/lib_${v}_prof.so timeout 300 python tools/w96_prof.py 64 2>&1 | tail -2 | tee -a $OUT/w96_prof.txt; fi
  if [ -f $V/lib_$v.so ]; then for d in default rings64; do echo "== stages 128 $d: $v"; SA3D_LIB=$V/lib_$v.so timeout 300 python tools/stages_at.py 128 data=$d 2>&1 | grep "m=256 16:259" | tee -a $OUT/stages.txt; done; fi
done
echo "== product"; timeout 300 python tools/w96_prof.py 64 2>&1 | tail -2 | tee -a $OUT/w96_prof.txt
for d in default rings64; do timeout 300 python tools/stages_at.py 128 data=$d 2>&1 | grep "m=256 16:259" | tee -a $OUT/stages.txt; done
echo "== mlp tests"; timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "mlp or dense or vote" -p no:cacheprovider -x > $OUT/pytest_mlp.log 2>&1; tail -2 $OUT/pytest_mlp.log
