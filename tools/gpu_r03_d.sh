#!/bin/bash
TAG=${1:-r03d}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests -m gpu -q -x -k "fps or dist or backbone or pin or pipeline or properties" -p no:cacheprovider -rf > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
timeout 300 python bench.py --steps 8 --warmup 2 --streams 1 --no-cpu-baseline --profile-iters 5 --verify 2 > $OUT/b1.json 2> $OUT/b1.err
python - <<P
import json
d = json.loads(open("$OUT/b1.json").read().strip().splitlines()[-1])
print("lat", d["single_stream_batch_latency_ms"], "verify", d["verify"]["all_equal_eager"])
for s in d["stages"]:
    if "fps" in s["kernel"] or "square" in s["kernel"]: print(s["kernel"], s["label"], s["avg_ms"])
P
timeout 300 python bench.py --no-cpu-baseline --profile-iters 0 --verify 0 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['value'], d['ms_per_step'])"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-iters 0 --verify 0 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('20 steps', d['value'], d['ms_per_step'])"
