export SA3D_LIB=3dssd_amd/csrc/variants/lib_sqk.so TMPDIR=/tmp
Q="--allow-knobs --no-cpu-baseline --no-other-executor --profile-iters 0 --verify 0 --steps 20 --warmup 5"
one() { python bench.py $Q "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['config']['timed_window_ms'], d['config']['probe_window_ms'])"; }
for rep in 1 2 3 4; do
echo "old";  SA_SQDIST_PERSIST=0 one
echo "persist3"; one
done
