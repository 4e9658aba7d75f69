#!/bin/bash
# throughput A/B of library variants: default (128 steps) and 20-step runs, REPS times each, interleaved
export TMPDIR=/tmp PYTHONUNBUFFERED=1
VARIANTS="$@"
for rep in $(seq 1 ${REPS:-3}); do for v in $VARIANTS; do
  if [ "$v" = "base" ]; then unset SA3D_LIB; K=""; else export SA3D_LIB=$GRAFT_REPO_ROOT/3dssd_amd/csrc/variants/lib_$v.so; K="--allow-knobs"; fi
  for st in "128:24" "20:5"; do S=${st%%:*}; W=${st##*:}
    timeout 300 python bench.py --steps $S --warmup $W --no-cpu-baseline --no-uncoalesced --profile-iters 0 --verify 0 $K 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v steps $S:', d['value'], d['ms_per_step'])"
  done
done; done
