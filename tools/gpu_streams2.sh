#!/bin/bash
# throughput vs number of streams / hardware queues
export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
for cfg in "16 16" "24 24" "32 32" "32 16" "12 12" "8 8"; do set -- $cfg
  GPU_MAX_HW_QUEUES=$2 timeout 300 python bench.py --steps 96 --warmup 16 --streams $1 --no-cpu-baseline --profile-iters 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams $1 queues $2 value', d['value'], 'ms/step', d['ms_per_step'])"
done
