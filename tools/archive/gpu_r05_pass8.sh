#!/bin/bash
# round 5, GPU pass 8: main streams x package size of the staged executor (512 and 20 steps), after the kernel changes
export TMPDIR=/tmp PYTHONUNBUFFERED=1
Q="--no-cpu-baseline --no-other-executor --profile-iters 0 --verify 0"
for cfg in "2 4 16" "3 6 16" "4 8 16" "3 6 8" "4 8 8" "2 4 8" "3 5 16"; do set -- $cfg
  a=$(timeout 300 python bench.py --main-streams $1 --streams $2 --coalesce $3 $Q | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])")
  b=$(timeout 300 python bench.py --main-streams $1 --streams $2 --coalesce $3 --steps 20 --warmup 5 $Q | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])")
  echo "main_streams=$1 packages=$2 batches_per_package=$3: 512 steps $a   20 steps $b"
done
echo "== done"
