#!/bin/bash
# round 3, first GPU pass: whole GPU suite, then the bench lines (default / 20 steps / data variants / 2 ranks on one device)
TAG=${1:-r03a}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider -rf > $OUT/pytest_gpu.log 2>&1; tail -15 $OUT/pytest_gpu.log
echo "== bench default"; timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -3 $OUT/bench_default.err
python - <<P
import json
def show(tag, f):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(tag, d["value"], d["unit"], "ms/step", d["ms_per_step"], "lat", d.get("single_stream_batch_latency_ms"), "verify", (d.get("verify") or {}).get("all_equal_eager"),
              "rows", (d.get("mlp_rows_per_step") or {}).get("evaluated_frac"), "fps_eval_frac", (d.get("roofline") or {}).get("evaluated_frac"))
    except Exception as e:
        print(tag, "failed", e)
show("default", "$OUT/bench_default.json")
P
echo "== bench 20 steps"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_20steps.json 2> $OUT/bench_20steps.err
for v in dup10 dense; do timeout 600 python bench.py --data $v --no-cpu-baseline > $OUT/bench_$v.json 2> $OUT/bench_$v.err; tail -2 $OUT/bench_$v.err; done
echo "== 2 ranks on one device"; timeout 600 python bench.py --gpus 2 --allow-shared-device --steps 64 --warmup 16 --no-cpu-baseline > $OUT/bench_2ranks_shared.json 2> $OUT/bench_2ranks_shared.err; tail -3 $OUT/bench_2ranks_shared.err
python - <<P
import json
for tag in ("20steps", "dup10", "dense", "2ranks_shared"):
    try:
        d = json.loads(open("$OUT/bench_%s.json" % tag).read().strip().splitlines()[-1])
        print(tag, d["value"], "ms/step", d["ms_per_step"], "lat", d.get("single_stream_batch_latency_ms"), "n_gpus", d["n_gpus"], "verify", (d.get("verify") or {}).get("all_equal_eager"),
              "rows", (d.get("mlp_rows_per_step") or {}).get("evaluated_frac"), "fps_eval_frac", (d.get("roofline") or {}).get("evaluated_frac"))
    except Exception as e:
        print(tag, "failed", e)
P
echo "== done"
