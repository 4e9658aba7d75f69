#!/bin/bash
# throughput A/B of library variants, 128-step runs only, REPS times each, interleaved
export TMPDIR=/tmp PYTHONUNBUFFERED=1
for rep in $(seq 1 ${REPS:-3}); do for v in "$@"; do
  if [ "$v" = "base" ]; then unset SA3D_LIB; K=""; else export SA3D_LIB=$GRAFT_REPO_ROOT/3dssd_amd/csrc/variants/lib_$v.so; K="--allow-knobs"; fi
  timeout 300 python bench.py --steps 128 --warmup 24 --no-cpu-baseline --no-uncoalesced --profile-iters 0 --verify 0 $K ${BENCH_EXTRA} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v:', d['value'], d['ms_per_step'])"
done; done
