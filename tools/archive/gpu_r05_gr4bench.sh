export TMPDIR=/tmp
Q="--no-cpu-baseline --no-other-executor --profile-iters 0 --verify 8"
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['verify']['all_equal_eager'])"; }
for rep in 1 2; do
python bench.py $Q | show "8-row default"
python tools/archive/bench_with.py MLP_GRANULE4=True -- $Q | show "4-row default"
done
python bench.py $Q --steps 20 --warmup 5 | show "8-row 20"
python tools/archive/bench_with.py MLP_GRANULE4=True -- $Q --steps 20 --warmup 5 | show "4-row 20"
python bench.py $Q --data rings64 | show "8-row rings64"
python tools/archive/bench_with.py MLP_GRANULE4=True -- $Q --data rings64 | show "4-row rings64"
