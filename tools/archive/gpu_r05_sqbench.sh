export SA3D_LIB=3dssd_amd/csrc/variants/lib_sqk.so TMPDIR=/tmp
Q="--allow-knobs --no-cpu-baseline --no-other-executor --profile-iters 0 --verify 0"
one() { python bench.py $Q "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
echo "old";  SA_SQDIST_PERSIST=0 one
echo "persist wgs3"; one
echo "persist wgs2"; SA_SQP_WGS=2 one
echo "persist wgs4"; SA_SQP_WGS=4 one
done
echo "20 steps old"; SA_SQDIST_PERSIST=0 one --steps 20 --warmup 5
echo "20 steps persist3"; one --steps 20 --warmup 5
echo "20 steps persist2"; SA_SQP_WGS=2 one --steps 20 --warmup 5
