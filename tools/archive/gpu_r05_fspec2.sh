export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_ref_pin_gpu.py tests/test_backbone_gpu.py -m gpu -q -x -k "fps or sampl or with_distance or backbone" -p no:cacheprovider 2>&1 | tail -2
for rep in 1 2; do
echo "== new"; python tools/stages_at.py 128 | grep -i "fps_dual\|non-FPS"
echo "== base"; SA3D_LIB=3dssd_amd/csrc/variants/lib_base.so python tools/stages_at.py 128 | grep -i "fps_dual\|non-FPS"
done
Q="--no-cpu-baseline --no-other-executor --profile-iters 0 --verify 8"
for rep in 1 2; do
python bench.py $Q | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new default', d['value'], d['verify']['all_equal_eager'])"
SA3D_LIB=3dssd_amd/csrc/variants/lib_base.so python bench.py --allow-knobs $Q | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base default', d['value'])"
python bench.py $Q --steps 20 --warmup 5 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new 20', d['value'])"
SA3D_LIB=3dssd_amd/csrc/variants/lib_base.so python bench.py --allow-knobs $Q --steps 20 --warmup 5 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base 20', d['value'])"
done
python bench.py $Q --workload configs2 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new configs2', d['value'])"
SA3D_LIB=3dssd_amd/csrc/variants/lib_base.so python bench.py --allow-knobs $Q --workload configs2 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base configs2', d['value'])"
