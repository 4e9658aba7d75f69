#!/bin/bash
# throughput over (slots, batches per package): $@ = "slots:coalesce" pairs; 512 steps of 8 frames each, then the driver's 20
# EXECUTOR=staged|slots (default staged)
export TMPDIR=/tmp PYTHONUNBUFFERED=1
for sc in "$@"; do S=${sc%%:*}; C=${sc##*:}
  for st in "512:64" "20:5"; do K=${st%%:*}; W=${st##*:}
    timeout 300 python bench.py --executor ${EXECUTOR:-staged} --streams $S --coalesce $C --steps $K --warmup $W --no-cpu-baseline --no-other-executor --profile-iters 0 --verify 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${EXECUTOR:-staged} slots $S coalesce $C steps $K:', d['value'], d['ms_per_step'], 'smallest package alone', d['single_stream_batch_latency_ms'])"
  done
done
