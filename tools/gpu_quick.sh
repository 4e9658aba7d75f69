#!/bin/bash
# quick GPU iteration: tests + a few bench variants.  usage: gpurun -- 'bash tools/gpu_quick.sh tag [pytest-args]'
TAG=${1:-q}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider -rf "$@" > $OUT/pytest_gpu.log 2>&1; tail -15 $OUT/pytest_gpu.log
for cfg in "1 0" "8 0" "8 1" "16 1"; do set -- $cfg
  timeout 300 python bench.py --steps 32 --warmup 8 --streams $1 --graphs $2 --no-cpu-baseline > $OUT/bench_s$1_g$2.json 2> $OUT/bench_s$1_g$2.err
  python - $OUT/bench_s$1_g$2.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("streams", d["config"]["streams"], "graphs", d["hip_graphs"], "value", d["value"], "ms/step", d["ms_per_step"], "lat", d["single_stream_batch_latency_ms"], "host", d["host_issue_ms_per_step"])
except Exception as e:
    print("bench failed", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
done
python - $OUT/bench_s1_g0.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
tot = 0
for s in d["stages"]:
    t = s["avg_ms"] * s["calls_per_step"]; tot += t
    print("   %-55s x%d %8.4f ms  %8.2f TF  %8.1f GB/s" % (s["label"], s["calls_per_step"], s["avg_ms"], s.get("tflops", 0), s.get("gbs", 0)))
print(" sum of kernels per step: %.3f ms" % tot, "mlp", d["roofline_grouped_mlp"]["achieved"], "TF")
PY
