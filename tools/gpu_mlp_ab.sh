#!/bin/bash
# A/B of MLP paths via env switches on the in-tree library
TAG=${1:-m}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
for cfg in "base:" "wide:SA_MLP_WIDE=1" "narrow8:SA_MLP_NARROW_NT=8" "narrow0:SA_MLP_NARROW_NT=0"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  echo "=== $name ($envs)"
  env $envs timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "mlp or dense" 2>&1 | tail -1
  for rep in 1 2; do
  env $envs timeout 300 python bench.py --steps 6 --warmup 2 --streams 1 --no-cpu-baseline --profile-iters 3 > $OUT/$name.$rep.json 2> $OUT/$name.$rep.err
  python - $OUT/$name.$rep.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
tot = 0; parts = []
for s in d["stages"]:
    if "group_mlp_max" in s["label"]:
        t = s["avg_ms"] * s["calls_per_step"]; tot += t
        parts.append("%s=%.3f" % (s["label"].replace("group_mlp_max ", "").replace(" ", ""), s["avg_ms"]))
print(" mlp total %.3f ms | lat %.3f |" % (tot, d["single_stream_batch_latency_ms"]), " ".join(parts))
PY
  done
done
