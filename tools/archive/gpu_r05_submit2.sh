export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q -x -k "pipeline" -p no:cacheprovider 2>&1 | tail -2
Q="--no-cpu-baseline --no-other-executor --profile-iters 0 --verify 32"
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 $Q | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('20 steps', d['value'], 'window', c['timed_window_ms'], 'probe', c['probe_window_ms'], 'issue', c['host_issue_total_ms'], c['host_issue_top_sites'], 'pkg0_reached', c['pkg0_reached_ms'], d['verify']['all_equal_eager'])"; done
python bench.py $Q | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['value'], d['ms_per_step'], d['verify']['all_equal_eager'])"
python bench.py --host-input $Q | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('host-input', d['value'], d['ms_per_step'], d['verify']['all_equal_eager'])"
python bench.py --graphs 0 --steps 20 --warmup 5 $Q | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('eager 20', d['value'], d['verify']['all_equal_eager'])"
