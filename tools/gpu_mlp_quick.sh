#!/bin/bash
# MLP parity tests + single-stream per-kernel times, row-wave kernel on and off
TAG=${1:-m}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_backbone_gpu.py -m gpu -q -p no:cacheprovider -k "mlp or backbone or config0" 2>&1 | tail -8
for cfg in "rowwave:SA_MLP_ROWWAVE=1" "generic:SA_MLP_ROWWAVE=0"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout 300 python bench.py --steps 8 --warmup 2 --streams 1 --no-cpu-baseline --profile-iters 3 > $OUT/$name.json 2> $OUT/$name.err
  python - $OUT/$name.json $name <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
tot = 0; parts = []
for s in d["stages"]:
    if "group_mlp_max" in s["label"]:
        tot += s["avg_ms"] * s["calls_per_step"]
        parts.append("%s=%.3f" % (s["label"].replace("group_mlp_max ", "").replace(" ", ""), s["avg_ms"]))
print(sys.argv[2], "mlp total %.3f ms | lat %.3f |" % (tot, d["single_stream_batch_latency_ms"]), " ".join(parts))
PY
done
env timeout 300 python bench.py --steps 32 --warmup 8 --no-cpu-baseline > $OUT/bench16.json 2> $OUT/bench16.err; python -c "
import json,sys; d=json.loads(open('$OUT/bench16.json').read().strip().splitlines()[-1]); print('16 streams value', d['value'], 'ms/step', d['ms_per_step'])"
