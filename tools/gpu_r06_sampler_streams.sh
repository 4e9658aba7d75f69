cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
Q="--no-cpu-baseline --no-other-executor --extras-budget 0 --profile-iters 0"
for ss in 1 2 1 2; do
  python bench.py --steps 20 --warmup 5 $Q --sampler-streams $ss 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('20 steps  sampler-streams $ss', d['value'], d['ms_per_step'], d['config'].get('timed_window_ms'))"
done
for ss in 1 2; do
  python bench.py $Q --sampler-streams $ss 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('512 steps sampler-streams $ss', d['value'], d['ms_per_step'])"
done
