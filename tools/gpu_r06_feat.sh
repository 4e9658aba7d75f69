#!/bin/bash
# round 6: the new host-side pieces -- detector through the executor, RCCL smoke, opt-in grid sharing, bench extras
TAG=${1:-r06_feat}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "== tests"; timeout 1200 python -m pytest tests/test_head.py tests/test_pipeline_gpu.py tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -x -k "head or detector or rccl or share_the_grid or postprocessor or nms or pipeline" > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
echo "== bench 20 steps with extras"; ( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_20steps.json 2> $OUT/bench_20steps.err ) 2>&1 | grep real
python - <<P
import json
d = json.loads(open("$OUT/bench_20steps.json").read().strip().splitlines()[-1]); c = d["config"]
print("value", d["value"], "ms/step", d["ms_per_step"])
print("first keys:", list(c)[:22])
print({k: c[k] for k in list(c)[7:19]})
print("extras:", json.dumps(d.get("extras"), indent=None)[:3000])
P
tail -5 $OUT/bench_20steps.err
echo "== bench detector"; timeout 600 python bench.py --workload detector --no-cpu-baseline --no-other-executor > $OUT/bench_detector.json 2> $OUT/bench_detector.err; python - <<P
import json
d = json.loads(open("$OUT/bench_detector.json").read().strip().splitlines()[-1])
print(d["metric"], d["value"], d["ms_per_step"], d["verify"], d.get("detections"))
print([ (s["kernel"], s["label"], s["avg_ms"]) for s in d["stages"][-10:]])
P
tail -3 $OUT/bench_detector.err
