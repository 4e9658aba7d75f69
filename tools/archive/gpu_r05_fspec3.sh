export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_ref_pin_gpu.py tests/test_backbone_gpu.py tests/test_pipeline_gpu.py -m gpu -q -x -k "fps or sampl or with_distance or backbone or forty or coalescing" -p no:cacheprovider 2>&1 | tail -2
for rep in 1 2; do
echo "== new"; python tools/stages_at.py 128 | grep -i "fps_dual\|non-FPS"
echo "== base"; SA3D_LIB=3dssd_amd/csrc/variants/lib_base.so python tools/stages_at.py 128 | grep -i "fps_dual\|non-FPS"
done
