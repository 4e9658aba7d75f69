#!/bin/bash
# round 5, GPU pass 2: stall experiments (quota / per-process SMI / sleeping waits), ball-query parity + timings on the
# three data variants with the pipelined walk, phase clocks of the layer-1 D-FPS kernel.
OUT=gpurun_out/r05_pass2; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
bash tools/gpu_r05_stall2.sh 2>&1 | tee $OUT/stall2.txt
echo "== ball query / backbone / pipeline tests"
timeout 900 python -m pytest tests -m gpu -q -x -k "ball or query or backbone or pipeline or ref_pin or fuzz" -p no:cacheprovider -rf > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
for d in default rings64; do echo "== stages at 128 frames, data=$d"; timeout 300 python tools/stages_at.py 128 data=$d 2>&1 | grep -v amdgpu.ids | tee $OUT/stages_128_$d.txt | grep -i "ball\|total"; done
echo "== stages at 32 frames, data=dense"; timeout 300 python tools/stages_at.py 32 data=dense 2>&1 | grep -v amdgpu.ids | tee $OUT/stages_32_dense.txt | grep -i "ball\|total"
echo "== D-FPS phase clocks"; bash tools/gpu_fpsb_prof.sh 2>&1 | grep -v amdgpu.ids | tee $OUT/fpsb_prof.txt
echo "== done"
