#!/bin/bash
TAG=${1:-s}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
run() { # hwq streams graphs
  GPU_MAX_HW_QUEUES=$1 timeout 300 python bench.py --steps 48 --warmup 8 --streams $2 --graphs $3 --no-cpu-baseline --profile-iters 1 > $OUT/b_$1_$2_$3.json 2> $OUT/b_$1_$2_$3.err
  python - $OUT/b_$1_$2_$3.json $1 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("hwq", sys.argv[2], "streams", d["config"]["streams"], "graphs", d["hip_graphs"], "value", d["value"], "ms/step", d["ms_per_step"], "host", d["host_issue_ms_per_step"])
except Exception as e:
    print("failed", e)
PY
}
for q in 4 8 16 24; do for s in 4 8 16 24; do run $q $s 0; done; done
run 16 16 1; run 24 24 1
