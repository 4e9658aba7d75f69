#!/bin/bash
# round 6: variants of the 96-row kernel -- parity, phase clocks at 64 frames and the layer-4 call at 128 frames
TAG=${1:-r06_w96b}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
V=$GRAFT_REPO_ROOT/3dssd_amd/csrc/variants
echo "== mlp tests"; timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "mlp or dense or vote" -p no:cacheprovider -x > $OUT/pytest_mlp.log 2>&1; tail -3 $OUT/pytest_mlp.log
for v in oldprof w96prof nopair_prof; do
  echo "== phase clocks $v"; SA3D_LIB=$V/lib_$v.so timeout 300 python tools/w96_prof.py 64 2>&1 | tail -2 | tee -a $OUT/w96_prof.txt
done
for v in r05base nopair; do
  echo "== stages 128 default: $v"; SA3D_LIB=$V/lib_$v.so timeout 300 python tools/stages_at.py 128 2>&1 | grep "m=256 16:259" | tee -a $OUT/stages.txt
done
echo "== stages 128 default: product"; timeout 300 python tools/stages_at.py 128 2>&1 | grep "m=256 16:259" | tee -a $OUT/stages.txt
echo "== stages 128 rings64: r05base"; SA3D_LIB=$V/lib_r05base.so timeout 300 python tools/stages_at.py 128 data=rings64 2>&1 | grep "m=256 16:259" | tee -a $OUT/stages.txt
echo "== stages 128 rings64: product"; timeout 300 python tools/stages_at.py 128 data=rings64 2>&1 | grep "m=256 16:259" | tee -a $OUT/stages.txt
echo "== done"
