#!/bin/bash
# round 3: MLP GEMM chain bring-up: parity tests of the MLP, then stage timings
TAG=${1:-r03b}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "group_mlp or backbone" -p no:cacheprovider -rf -s > $OUT/pytest_mlp.log 2>&1; grep -E "gemm chain|passed|failed|Error|error" $OUT/pytest_mlp.log | tail -20
timeout 600 python bench.py --steps 64 --warmup 16 --no-cpu-baseline --verify 16 > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err
python - <<P
import json
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print(d["value"], "ms/step", d["ms_per_step"], "lat", d["single_stream_batch_latency_ms"], "verify", d["verify"]["all_equal_eager"])
for s in d["stages"]:
    if "mlp" in s["kernel"] or "dense" in s["kernel"]: print(s["kernel"], s["label"][:70], s["avg_ms"])
print(json.dumps(d["roofline_grouped_mlp"])[:400])
P
