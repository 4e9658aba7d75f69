export TMPDIR=/tmp
Q="--no-cpu-baseline --no-other-executor --profile-iters 0 --verify 8"
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['verify']['all_equal_eager'])"; }
for v in default rings64 dense; do
for rep in 1 2; do
python bench.py $Q --data $v | show "8-row $v"
python tools/archive/bench_with.py "MLP_GRANULE4={1024}" -- $Q --data $v | show "layer-2 4-row $v"
done
done
python bench.py $Q --steps 20 --warmup 5 | show "8-row 20"
python tools/archive/bench_with.py "MLP_GRANULE4={1024}" -- $Q --steps 20 --warmup 5 | show "layer-2 4-row 20"
