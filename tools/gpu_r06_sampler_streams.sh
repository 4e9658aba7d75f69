#!/bin/bash
# one or two sampler streams (stage A of consecutive packages side by side), with ROCm's default four hardware queues and with six
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
Q="--no-cpu-baseline --no-other-executor --extras-budget 0 --profile-iters 0"
for cfg in "--sampler-streams 1" "--sampler-streams 2" "--sampler-streams 2 --hw-queues 6" "--sampler-streams 1 --hw-queues 6"; do
  for rep in 1 2; do
    python bench.py --steps 20 --warmup 5 $Q $cfg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('20 steps  $cfg:', d['value'], d['ms_per_step'], d['config'].get('timed_window_ms'))"
  done
  python bench.py $Q $cfg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('512 steps $cfg:', d['value'], d['ms_per_step'])"
done
