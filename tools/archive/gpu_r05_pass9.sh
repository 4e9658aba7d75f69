#!/bin/bash
# round 5, GPU pass 9: readlane ranking in the grid ball query -- parity and timings on the three data variants
OUT=gpurun_out/r05_pass9; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q -x -k "ball or query or backbone or ref_pin or fuzz or pipeline" -p no:cacheprovider -rf > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for d in default rings64; do echo "== stages at 128 frames, data=$d"; timeout 300 python tools/stages_at.py 128 data=$d 2>&1 | grep -i "ball\|total"; done
echo "== stages at 32 frames, data=dense"; timeout 300 python tools/stages_at.py 32 data=dense 2>&1 | grep -i "ball\|total"
echo "== done"
