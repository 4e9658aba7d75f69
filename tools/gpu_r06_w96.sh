#!/bin/bash
# round 6: the 96-row kernel with its phases overlapped through LDS -- parity, phase clocks, stage A/B against the round-5 library
TAG=${1:-r06_w96}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
V=$GRAFT_REPO_ROOT/3dssd_amd/csrc/variants
echo "== mlp tests"; timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "mlp or dense or vote" -p no:cacheprovider -x > $OUT/pytest_mlp.log 2>&1; tail -3 $OUT/pytest_mlp.log
echo "== backbone tests"; timeout 900 python -m pytest tests/test_backbone_gpu.py tests/test_pipeline_gpu.py -m gpu -q -p no:cacheprovider -x > $OUT/pytest_bb.log 2>&1; tail -3 $OUT/pytest_bb.log
echo "== phase clocks (new)"; timeout 300 python tools/w96_prof.py 2>&1 | tail -2 | tee $OUT/w96_prof.txt
for d in default rings64; do
  echo "== stages 128 $d: round-5 library"; SA3D_LIB=$V/lib_r05base.so timeout 300 python tools/stages_at.py 128 data=$d 2>&1 | grep "group_mlp_max_layer\|total" | tee $OUT/stages_base_$d.txt
  echo "== stages 128 $d: new";             timeout 300 python tools/stages_at.py 128 data=$d 2>&1 | grep "group_mlp_max_layer\|total" | tee $OUT/stages_new_$d.txt
done
echo "== done"
