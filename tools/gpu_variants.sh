#!/bin/bash
# A/B of library builds: for every 3dssd_amd/csrc/variants/lib_*.so run the MLP parity tests and print per-kernel times
TAG=${1:-v}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1 GPU_MAX_HW_QUEUES=16
for lib in 3dssd_amd/csrc/variants/lib_*.so; do
  name=$(basename $lib .so)
  export SA3D_LIB=$PWD/$lib
  echo "=== $name"
  timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "${PYTEST_K:-mlp or dense}" 2>&1 | tail -2
  for rep in 1 2; do
  timeout 300 python bench.py --steps 6 --warmup 2 --streams 1 --no-cpu-baseline --profile-iters 3 > $OUT/$name.$rep.json 2> $OUT/$name.$rep.err
  python - $OUT/$name.$rep.json "${STAGE_FILTER:-group_mlp_max dense}" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
filt = sys.argv[2].split()
tot = 0; parts = []
for s in d["stages"]:
    if any(f in s["label"] for f in filt):
        t = s["avg_ms"] * s["calls_per_step"]; tot += t
        parts.append("%s=%.3f" % (s["label"].replace("group_mlp_max ", "").replace(" ", ""), s["avg_ms"]))
print(" total %.3f ms | lat %.3f |" % (tot, d["single_stream_batch_latency_ms"]), " ".join(parts))
PY
  done
done
