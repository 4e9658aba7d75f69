#!/bin/bash
# A/B of env switches on the in-tree library: usage gpu_env_ab.sh tag "name:ENV=..;name2:ENV2=.." [stage-filter]
TAG=${1:-e}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
IFS=';' read -ra CFGS <<< "$2"
for cfg in "${CFGS[@]}"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout 300 python bench.py --steps 8 --warmup 2 --streams 1 --no-cpu-baseline --profile-iters 3 > $OUT/$name.json 2> $OUT/$name.err
  python - $OUT/$name.json $name "${3:-group_mlp_max}" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
tot = 0; parts = []
for s in d["stages"]:
    if sys.argv[3] in s["label"]:
        tot += s["avg_ms"] * s["calls_per_step"]
        parts.append("%s=%.3f" % (s["label"].replace("group_mlp_max ", "").replace(" ", ""), s["avg_ms"]))
print(sys.argv[2], "total %.3f ms | lat %.3f |" % (tot, d["single_stream_batch_latency_ms"]), " ".join(parts))
PY
done
