#!/bin/bash
# A/B of the D-FPS kernels: parity tests, then the single-stream bench with and without the wave-bucket kernel
TAG=${1:-f}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "fps" 2>&1 | tail -5
for cfg in "plain:SA_FPS_BUCKET_MIN_N=0" "bucket:SA_FPS_BUCKET_MIN_N=8192"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout 300 python bench.py --steps 8 --warmup 2 --streams 1 --no-cpu-baseline --profile-iters 3 > $OUT/$name.json 2> $OUT/$name.err
  python - $OUT/$name.json $name <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
parts = ["%s=%.3f" % (s["label"].replace(" ", ""), s["avg_ms"]) for s in d["stages"] if "fps" in s["label"]]
print(sys.argv[2], "lat %.3f |" % d["single_stream_batch_latency_ms"], " ".join(parts))
PY
done
