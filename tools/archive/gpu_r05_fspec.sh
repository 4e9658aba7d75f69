export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_ref_pin_gpu.py -m gpu -q -x -k "fps or sampl or with_distance" -p no:cacheprovider 2>&1 | tail -2
SA3D_LIB=3dssd_amd/csrc/variants/lib_fspec.so timeout 200 python tools/archive/ffps_spec_stats.py default rings64 dense
for rep in 1 2; do
echo "== new"; python tools/stages_at.py 128 | grep -i "fps_dual"
echo "== base"; SA3D_LIB=3dssd_amd/csrc/variants/lib_base.so python tools/stages_at.py 128 | grep -i "fps_dual"
done
echo "== rings64 new"; python tools/stages_at.py 128 data=rings64 | grep -i "fps_dual"
echo "== rings64 base"; SA3D_LIB=3dssd_amd/csrc/variants/lib_base.so python tools/stages_at.py 128 data=rings64 | grep -i "fps_dual"
