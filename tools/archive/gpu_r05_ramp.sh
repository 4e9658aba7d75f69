export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_pipeline_gpu.py tests/test_bench_contract.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
Q="--no-cpu-baseline --no-other-executor --profile-iters 0 --verify 32"
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('$1', d['value'], c['timed_window_ms'], c['probe_window_ms'], d['verify']['all_equal_eager'], c.get('timed_packages_ms',{}).get('rows'))"; }
for rep in 1 2 3; do
python bench.py $Q --steps 20 --warmup 5 | show "ramp 20"
python bench.py $Q --steps 20 --warmup 5 --ramp-stream 0 | show "none 20"
done
for rep in 1 2; do
python bench.py $Q | show "ramp 512"
python bench.py $Q --ramp-stream 0 | show "none 512"
done
